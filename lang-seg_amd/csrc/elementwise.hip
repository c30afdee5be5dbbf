// elementwise.hip -- the HBM-bound kernels of the LSeg forward (gfx950): LayerNorm, im2col,
// readout concat, bilinear x2 (NHWC bf16 and NCHW fp32 planes), the L2-norm/scale/fp16 cast
// in front of the correlation GEMM, text embedding / pooling / normalisation, one-off weight
// repacking.  All are coalesced 16-byte-per-lane streams (cdna_hip_programming.md G2/G13).
#include "ops.h"
#include "../../include/lseg_hip.h"

namespace lseg {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- LayerNorm: one wave per row; row cached in registers (D <= 64*4*MAXV) --------------------------
// [3P] timm norm1/norm2 (eps 1e-6, fp32 in) and CLIP's fp32-computing LayerNorm (eps 1e-5, fp16 in).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* in, int in_dtype, const float* gamma,
                                                        const float* beta, void* out, int out_dtype,
                                                        int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;                      // float4 groups per row
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            if (in_dtype == DT_F32) {
                v[i] = reinterpret_cast<const float4*>((const float*)in + (size_t)row * D)[g];
            } else {
                const uint2 u = reinterpret_cast<const uint2*>((const uint16_t*)in + (size_t)row * D)[g];
                v[i].x = load_as_f32(&u, 0, in_dtype); v[i].y = load_as_f32(&u, 1, in_dtype);
                v[i].z = load_as_f32(&u, 2, in_dtype); v[i].w = load_as_f32(&u, 3, in_dtype);
            }
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            const float4 ga = reinterpret_cast<const float4*>(gamma)[g];
            const float4 be = reinterpret_cast<const float4*>(beta)[g];
            float y[4] = {(v[i].x - mean) * rstd * ga.x + be.x, (v[i].y - mean) * rstd * ga.y + be.y,
                          (v[i].z - mean) * rstd * ga.z + be.z, (v[i].w - mean) * rstd * ga.w + be.w};
            if (out_dtype == DT_F32) {
                reinterpret_cast<float4*>((float*)out + (size_t)row * D)[g] = make_float4(y[0], y[1], y[2], y[3]);
            } else {
                uint2 pk;
                pk.x = pack2_dt(y[0], y[1], out_dtype);
                pk.y = pack2_dt(y[2], y[3], out_dtype);
                reinterpret_cast<uint2*>((uint16_t*)out + (size_t)row * D)[g] = pk;
            }
        }
    }
}

// ---- residual update + LayerNorm behind a split-K GEMM (small batches: engine.hip "split-K residual GEMMs") ----------------------------
// x[row] += bias + sum_s part[s * stride + row * D ...] (fixed order: deterministic), written back to the fp32 residual stream, then
// LayerNorm of the updated row (gamma == NULL: update only).  One wave per row like layernorm_kernel.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_reduce_kernel(float* __restrict__ x, const float* __restrict__ part, int nsplit, size_t stride,
                                                               const float* __restrict__ bias, const float* gamma, const float* beta,
                                                               void* out, int out_dtype, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    // All loads of a round are in flight together: x + bias, then the slabs FOUR at a time (16 independent 16-byte loads per lane).  The
    // round-3 form walked the slabs in a run-time loop inside the per-chunk loop -- ~20 dependent memory round trips per row, 10.7 us per
    // launch at B = 1 where the data volume is worth 3.  Same sums in the same order (x + bias + slab 0 + slab 1 + ...): same bits.
    float4 v[MAXV];
    int gi[MAXV];                                  // this lane's chunk index, clamped into the row (loads stay unconditional)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) gi[i] = lane + 64 * i < nv ? lane + 64 * i : nv - 1;
    {
        float4 b[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            v[i] = reinterpret_cast<const float4*>(x + (size_t)row * D)[gi[i]];
            b[i] = reinterpret_cast<const float4*>(bias)[gi[i]];
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) { v[i].x += b[i].x; v[i].y += b[i].y; v[i].z += b[i].z; v[i].w += b[i].w; }
    }
    for (int s0 = 0; s0 < nsplit; s0 += 4) {
        float4 p[4][MAXV];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int sp = s0 + c < nsplit ? s0 + c : nsplit - 1;
            const float4* src = reinterpret_cast<const float4*>(part + (size_t)sp * stride + (size_t)row * D);
#pragma unroll
            for (int i = 0; i < MAXV; ++i) p[c][i] = src[gi[i]];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool on = s0 + c < nsplit;               // wave-uniform
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                v[i].x += on ? p[c][i].x : 0.f; v[i].y += on ? p[c][i].y : 0.f;
                v[i].z += on ? p[c][i].z : 0.f; v[i].w += on ? p[c][i].w : 0.f;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            reinterpret_cast<float4*>(x + (size_t)row * D)[g] = v[i];
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (!gamma) return;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            const float4 ga = reinterpret_cast<const float4*>(gamma)[g];
            const float4 be = reinterpret_cast<const float4*>(beta)[g];
            const float y[4] = {(v[i].x - mean) * rstd * ga.x + be.x, (v[i].y - mean) * rstd * ga.y + be.y,
                                (v[i].z - mean) * rstd * ga.z + be.z, (v[i].w - mean) * rstd * ga.w + be.w};
            uint2 pk;
            pk.x = pack2_dt(y[0], y[1], out_dtype);
            pk.y = pack2_dt(y[2], y[3], out_dtype);
            reinterpret_cast<uint2*>((uint16_t*)out + (size_t)row * D)[g] = pk;
        }
    }
}

// ---- epilogue of a split-K 3x3 conv (engine.hip conv3x3, small batches): sum the fp32 partial slabs [ns][B*Ho*Wo][C], add the bias, optional
// ReLU, optional skip inputs (padded NHWC, like the output), write the padded-NHWC map and optionally its ReLU copy.  8 channels per thread.
__global__ void conv_reduce_pad_kernel(const float* __restrict__ part, int ns, size_t stride, const float* __restrict__ bias,
                                       const uint16_t* __restrict__ res, const uint16_t* __restrict__ res2, uint16_t* __restrict__ out,
                                       uint16_t* __restrict__ out_relu, int B, int Ho, int Wo, int C, int relu, int dtype) {
    const int c8n = C >> 3;
    const size_t total = (size_t)B * Ho * Wo * c8n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8;
        const size_t m = i / c8n;
        const int x = (int)(m % Wo), y = (int)((m / Wo) % Ho), b = (int)(m / ((size_t)Wo * Ho));
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bias ? bias[c + k] : 0.f;
        for (int sp = 0; sp < ns; ++sp) {
            const float4 a = *reinterpret_cast<const float4*>(part + (size_t)sp * stride + m * C + c);
            const float4 d = *reinterpret_cast<const float4*>(part + (size_t)sp * stride + m * C + c + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += d.x; v[5] += d.y; v[6] += d.z; v[7] += d.w;
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
        const size_t o = (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c;
        if (res) {
            const uint4 u = *reinterpret_cast<const uint4*>(res + o);
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&u);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += load_as_f32(e, k, dtype);
            if (res2) {
                const uint4 u2 = *reinterpret_cast<const uint4*>(res2 + o);
                const uint16_t* e2 = reinterpret_cast<const uint16_t*>(&u2);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += load_as_f32(e2, k, dtype);
            }
        }
        const uint4 pk = make_uint4(pack2_dt(v[0], v[1], dtype), pack2_dt(v[2], v[3], dtype), pack2_dt(v[4], v[5], dtype), pack2_dt(v[6], v[7], dtype));
        *reinterpret_cast<uint4*>(out + o) = pk;
        if (out_relu) {     // the same rounding first, then the sign test (identical to the GEMM epilogue's ReLU copy)
            auto rl = [](uint32_t w) { return ((w & 0x8000u) ? 0u : (w & 0xffffu)) | ((w & 0x80000000u) ? 0u : (w & 0xffff0000u)); };
            *reinterpret_cast<uint4*>(out_relu + o) = make_uint4(rl(pk.x), rl(pk.y), rl(pk.z), rl(pk.w));
        }
    }
}

// ---- patch im2col: x fp32 NCHW -> A [B*gh*gw, 3*P*P] (k = c*P*P + i*P + j), lseg_vit.py:179 ------
__global__ void im2col_patch_kernel(const float* x, uint16_t* A, int B, int H, int W, int P, int dtype) {
    const int gh = H / P, gw = W / P, Kd = 3 * P * P;
    const unsigned total = (unsigned)B * gh * gw * (Kd / 8);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int k8 = (int)(idx % (unsigned)(Kd / 8));
        const unsigned m = idx / (unsigned)(Kd / 8);
        const int k = k8 * 8;
        const int c = k / (P * P), rem = k - c * P * P, i = rem / P, j = rem - i * P;
        const int b = (int)(m / (unsigned)(gh * gw)), p = (int)(m - (unsigned)b * gh * gw), py = p / gw, px = p - py * gw;
        const float* src = x + (((size_t)b * 3 + c) * H + py * P + i) * W + px * P + j;
        const float4 f0 = reinterpret_cast<const float4*>(src)[0];
        const float4 f1 = reinterpret_cast<const float4*>(src)[1];
        const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pack2_dt(f[2 * e], f[2 * e + 1], dtype);
        }
        reinterpret_cast<uint4*>(A + (size_t)m * Kd + k)[0] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- pos-embed bilinear resize (align_corners=False), lseg_vit.py:149-163; run once per (H,W) ----
__global__ void pos_resize_kernel(const float* pos, float* out, int g_old, int gh, int gw, int D) {
    // pos [1 + g_old^2, D] -> out [1 + gh*gw, D]
    const size_t total = (size_t)(1 + gh * gw) * D;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % D);
        const int t = (int)(idx / D);
        if (t == 0) { out[idx] = pos[d]; continue; }
        const int y = (t - 1) / gw, x = (t - 1) - y * gw;
        // PyTorch upsample_bilinear2d, align_corners=False: src = max(0, (dst+0.5)*scale - 0.5)
        const float sy = fmaxf(0.f, ((float)y + 0.5f) * ((float)g_old / (float)gh) - 0.5f);
        const float sx = fmaxf(0.f, ((float)x + 0.5f) * ((float)g_old / (float)gw) - 0.5f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < g_old - 1), x1 = x0 + (x0 < g_old - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float* g = pos + D;   // grid part
        const float v00 = g[((size_t)y0 * g_old + x0) * D + d], v01 = g[((size_t)y0 * g_old + x1) * D + d];
        const float v10 = g[((size_t)y1 * g_old + x0) * D + d], v11 = g[((size_t)y1 * g_old + x1) * D + d];
        out[idx] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

// ---- cls token rows: x[b, 0, :] = cls + pos[0]  (lseg_vit.py:188-193) ---------------------------------
__global__ void cls_rows_kernel(const float* cls, const float* pos, float* x, int B, int ntok, int D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx - b * D;
    x[(size_t)b * ntok * D + d] = cls[d] + pos[d];
}

// ---- ProjectReadout concat (lseg_vit.py:87-88): x fp32 [B,N,D] -> A [B*(N-1), 2D] -----------------------
__global__ void readout_cat_kernel(const float* x, uint16_t* A, int B, int ntok, int D, int dtype) {
    const int g8 = 2 * D / 8;
    const unsigned total = (unsigned)B * (ntok - 1) * g8;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % (unsigned)g8);
        const unsigned m = idx / (unsigned)g8;
        const int b = (int)(m / (unsigned)(ntok - 1)), t = (int)(m - (unsigned)b * (ntok - 1));
        const int c = c8 * 8;
        const float* src = c < D ? x + ((size_t)b * ntok + t + 1) * D + c : x + (size_t)b * ntok * D + (c - D);
        const float4 f0 = reinterpret_cast<const float4*>(src)[0];
        const float4 f1 = reinterpret_cast<const float4*>(src)[1];
        const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pack2_dt(f[2 * e], f[2 * e + 1], dtype);
        }
        reinterpret_cast<uint4*>(A + (size_t)m * (2 * D) + c)[0] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- bilinear x2, align_corners=True, NHWC 16-bit: padded in [B,H+2,W+2,C] -> out [B,2H,2W,C] ------
// (FeatureFusionBlock_custom.forward, lseg_blocks.py:352-354)
__global__ void upsample2x_nhwc_kernel(const uint16_t* in, uint16_t* out, int B, int H, int W, int C, int dtype) {
    const int c8n = C / 8, Ho = 2 * H, Wo = 2 * W;
    const unsigned total = (unsigned)B * Ho * Wo * c8n;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % (unsigned)c8n);
        unsigned p = idx / (unsigned)c8n;
        const int xo = (int)(p % (unsigned)Wo); p /= (unsigned)Wo;
        const int yo = (int)(p % (unsigned)Ho);
        const int b = (int)(p / (unsigned)Ho);
        const float sy = ry * (float)yo, sx = rx * (float)xo;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const size_t rowp = (size_t)(W + 2) * C;
        const uint16_t* base = in + (size_t)b * (H + 2) * rowp + (size_t)c8 * 8;
        const uint4 a00 = *reinterpret_cast<const uint4*>(base + (size_t)(y0 + 1) * rowp + (size_t)(x0 + 1) * C);
        const uint4 a01 = *reinterpret_cast<const uint4*>(base + (size_t)(y0 + 1) * rowp + (size_t)(x1 + 1) * C);
        const uint4 a10 = *reinterpret_cast<const uint4*>(base + (size_t)(y1 + 1) * rowp + (size_t)(x0 + 1) * C);
        const uint4 a11 = *reinterpret_cast<const uint4*>(base + (size_t)(y1 + 1) * rowp + (size_t)(x1 + 1) * C);
        const uint32_t* p00 = reinterpret_cast<const uint32_t*>(&a00);
        const uint32_t* p01 = reinterpret_cast<const uint32_t*>(&a01);
        const uint32_t* p10 = reinterpret_cast<const uint32_t*>(&a10);
        const uint32_t* p11 = reinterpret_cast<const uint32_t*>(&a11);
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int sh = hlf * 16;
                const uint16_t u00 = (uint16_t)(p00[e] >> sh), u01 = (uint16_t)(p01[e] >> sh);
                const uint16_t u10 = (uint16_t)(p10[e] >> sh), u11 = (uint16_t)(p11[e] >> sh);
                const float v00 = load_as_f32(&u00, 0, dtype), v01 = load_as_f32(&u01, 0, dtype);
                const float v10 = load_as_f32(&u10, 0, dtype), v11 = load_as_f32(&u11, 0, dtype);
                r[hlf] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            }
            o[e] = pack2_dt(r[0], r[1], dtype);
        }
        *reinterpret_cast<uint4*>(out + (((size_t)b * Ho + yo) * Wo + xo) * C + (size_t)c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- bilinear x2, align_corners=True, fp32 planes [P,H,W] -> [P,2H,2W] (lseg_net.py:203) -----------
// optional per-plane post-op none.  Each thread writes 4 consecutive outputs (16 B).
__global__ __launch_bounds__(128) void upsample2x_planes_kernel(const float* in, float* out, int P, int H, int W) {
    // One block = one input-row pair -> the (up to 3) output rows whose source row y0 is this one.
    // The two input rows are staged in LDS (coalesced float4 loads); every thread then produces 4
    // consecutive outputs per output row from LDS and writes them with one 16-byte non-temporal
    // store.  Output row yo belongs to the block with y0 = floor(yo*(H-1)/(2H-1)).
    extern __shared__ float rows[];           // [2][W]
    const int Ho = 2 * H, Wo = 2 * W, w4 = Wo / 4;
    const unsigned pl = blockIdx.x / (unsigned)H;
    const int y0 = (int)(blockIdx.x - pl * H);
    const int y1 = y0 + (y0 < H - 1);
    const float* r0 = in + ((size_t)pl * H + y0) * W;
    const float* r1 = in + ((size_t)pl * H + y1) * W;
    for (int i = threadIdx.x; i < W / 4; i += 128) {
        reinterpret_cast<float4*>(rows)[i] = reinterpret_cast<const float4*>(r0)[i];
        reinterpret_cast<float4*>(rows + W)[i] = reinterpret_cast<const float4*>(r1)[i];
    }
    __syncthreads();
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    // first output row mapped to y0: smallest yo with floor(ry*yo) == y0
    int yo = (int)ceilf((float)y0 / ry);
    while (yo > 0 && (int)(ry * (float)(yo - 1)) >= y0) --yo;
    while ((int)(ry * (float)yo) < y0) ++yo;
    const int yo_first = yo;
    for (int x4 = threadIdx.x; x4 < w4; x4 += 128) {          // column terms once per thread, then down the (up to 3) output rows
        int x0[4], x1[4];
        float lx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) src_tap(rx, x4 * 4 + e, W, x0[e], x1[e], lx[e]);
        for (yo = yo_first; yo < Ho && (int)(ry * (float)yo) == y0; ++yo) {
            int ya_, yb_;
            float ly;
            src_tap(ry, yo, H, ya_, yb_, ly);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bilerp(rows[x0[e]], rows[x1[e]], rows[W + x0[e]], rows[W + x1[e]], lx[e], ly);
            const f32x4_t ov = {o[0], o[1], o[2], o[3]};     // written once, never re-read by the engine
            __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t*>(out + ((size_t)pl * Ho + yo) * Wo) + x4);
        }
    }
}

// ---- pixel feature normalise + scale + fp16 casts (lseg_net.py:191,194) ----------------------------------
// a[m,:] = fp16( scale * fp16( f[m,:] / ||f[m,:]||_2 ) )   (fp32 norm, two fp16 roundings).
template <int MAXV>
__global__ __launch_bounds__(256) void l2norm_scale_f16_kernel(const float* f, uint16_t* a, int M, int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            v[i] = reinterpret_cast<const float4*>(f + (size_t)row * C)[g];
            s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    const float nrm = sqrtf(wave_sum(s));
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            const float y[4] = {v[i].x / nrm, v[i].y / nrm, v[i].z / nrm, v[i].w / nrm};
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = f32_to_f16(scale * round_f16(y[e]));
            uint2 pk;
            pk.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
            pk.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
            reinterpret_cast<uint2*>(a + (size_t)row * C)[g] = pk;
        }
    }
}

// ---- commuted head: the pixel features without the 240x240 GEMMs ----------------------------------------------------------------
// scratch.head1 (1x1, lseg_net.py:185) follows refinenet1.out_conv (1x1, lseg_blocks.py:356), which follows the x2 bilinear upsample
// (:352-354).  1x1 convs commute with the (linear, per-channel) upsample, so
//     head1(out_conv(up(t))) = up(Wc t + bc),   Wc = Wh Wo,  bc = Wh bo + bh
// and the two GEMMs run ONCE at the quarter resolution (3.8 GF instead of 7.5 + 15.1 GF per image; exact in real arithmetic, fp32
// round-off apart).  combine_1x1_kernel builds (Wc, bc) in fp32 at pack time; upsample_norm_f16_kernel then produces the correlation's
// A operand straight from g = Wc t + bc (fp32, padded NHWC): bilinear x2 (align_corners=True), fp32 L2-norm over the channels and the
// two fp16 roundings of lseg_net.py:191,194 -- one wave per output pixel, 8 channels per lane per pass, the four taps read as float4.
__global__ void combine_1x1_kernel(const float* __restrict__ wh, const float* __restrict__ bh, const float* __restrict__ wo,
                                   const float* __restrict__ bo, float* __restrict__ wc, float* __restrict__ bc, int Co, int Cm, int Ci) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Co * (Ci + 1)) return;
    const int co = idx / (Ci + 1), ci = idx - co * (Ci + 1);
    double s = 0.0;
    if (ci < Ci) {
        for (int m = 0; m < Cm; ++m) s += (double)wh[(size_t)co * Cm + m] * (double)wo[(size_t)m * Ci + ci];
        wc[(size_t)co * Ci + ci] = (float)s;
    } else {
        for (int m = 0; m < Cm; ++m) s += (double)wh[(size_t)co * Cm + m] * (double)bo[m];
        bc[co] = (float)(s + (double)bh[co]);
    }
}
// One wave per 2x2 block of OUTPUT pixels, visited back to back: with align_corners=True and scale (H-1)/(2H-1) < 1/2 their 16 taps
// fall on a 3x3 input window, so 7 of the 16 tap reads (2 KB each at C = 512) are served by the CU's L1 instead of the L2 (a
// pixel-per-wave schedule re-read every tap from L2: 8 KB per output pixel, L2-bandwidth-bound at 1.4 ms for B = 36).
template <int NV>      // NV float4 pairs per lane: C = 512 -> NV = 1 (8 channels per lane), C = 768 -> 2 (lanes 0-31 only on the second)
__global__ __launch_bounds__(256, NV == 1 ? 3 : 2) void upsample_norm_f16_kernel(const float* __restrict__ g, uint16_t* __restrict__ a, int B, int H, int W,
                                                                int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t nblk = (size_t)B * H * W;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    const size_t rowp = (size_t)(W + 2) * C;
    for (size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += (size_t)gridDim.x * 4) {
        const int bx = (int)(blk % W);
        const int by = (int)((blk / W) % H);
        const int b = (int)(blk / ((size_t)W * H));
        const float* base = g + (size_t)b * (H + 2) * rowp;
#pragma unroll 1
        for (int dy = 0; dy < 2; ++dy) {
            // both pixels of the row pair are in flight together (the kernel is latency-bound: 16 independent 16-byte loads per lane)
            const int yo = 2 * by + dy;
            const float sy = ry * (float)yo;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1);
            const float ly = sy - (float)y0;
            float v[2][NV][8], lxs[2];
            const float* pr0 = base + (size_t)(y0 + 1) * rowp;
            const float* pr1 = base + (size_t)(y1 + 1) * rowp;
            float4 tp[2][NV][2][4];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float sx = rx * (float)(2 * bx + dx);
                const int x0 = (int)sx, x1 = x0 + (x0 < W - 1);
                lxs[dx] = sx - (float)x0;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = (lane + 64 * i) * 8;
                    if (c < C) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            tp[dx][i][h][0] = *reinterpret_cast<const float4*>(pr0 + (size_t)(x0 + 1) * C + c + 4 * h);
                            tp[dx][i][h][1] = *reinterpret_cast<const float4*>(pr0 + (size_t)(x1 + 1) * C + c + 4 * h);
                            tp[dx][i][h][2] = *reinterpret_cast<const float4*>(pr1 + (size_t)(x0 + 1) * C + c + 4 * h);
                            tp[dx][i][h][3] = *reinterpret_cast<const float4*>(pr1 + (size_t)(x1 + 1) * C + c + 4 * h);
                        }
                    }
                }
            }
            float s2[2] = {0.f, 0.f};
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float lx = lxs[dx];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if ((lane + 64 * i) * 8 < C) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float4 a00 = tp[dx][i][h][0], a01 = tp[dx][i][h][1], a10 = tp[dx][i][h][2], a11 = tp[dx][i][h][3];
                            // same association as upsample_bilinear2d: (1-ly)*((1-lx)*v00 + lx*v01) + ly*((1-lx)*v10 + lx*v11)
                            v[dx][i][4 * h + 0] = (1.f - ly) * ((1.f - lx) * a00.x + lx * a01.x) + ly * ((1.f - lx) * a10.x + lx * a11.x);
                            v[dx][i][4 * h + 1] = (1.f - ly) * ((1.f - lx) * a00.y + lx * a01.y) + ly * ((1.f - lx) * a10.y + lx * a11.y);
                            v[dx][i][4 * h + 2] = (1.f - ly) * ((1.f - lx) * a00.z + lx * a01.z) + ly * ((1.f - lx) * a10.z + lx * a11.z);
                            v[dx][i][4 * h + 3] = (1.f - ly) * ((1.f - lx) * a00.w + lx * a01.w) + ly * ((1.f - lx) * a10.w + lx * a11.w);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) s2[dx] += v[dx][i][e] * v[dx][i][e];
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s2[0] += __shfl_xor(s2[0], o); s2[1] += __shfl_xor(s2[1], o); }
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float nrm = sqrtf(s2[dx]);
                const size_t pix = ((size_t)b * Ho + yo) * Wo + 2 * bx + dx;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = (lane + 64 * i) * 8;
                    if (c < C) {
                        uint32_t o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t lo = f32_to_f16(scale * round_f16(v[dx][i][2 * e] / nrm)), hi = f32_to_f16(scale * round_f16(v[dx][i][2 * e + 1] / nrm));
                            o[e] = lo | (hi << 16);
                        }
                        *reinterpret_cast<uint4*>(a + pix * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
    }
}

// ---- commuted correlation (engine.hip "commuted correlation"): the pixel x text product at the QUARTER resolution ------------------
// logits_low[k, P] = t_k . a_P with a_P = s * u_P / ||u_P|| and u_P = the x2 bilinear of g at P is linear in u_P, so
//   t_k . u_P = sum_taps w_tap (t_k . g_tap)   -> R = T g^T is one GEMM at (H, W) instead of (2H, 2W) (4x fewer pixels), up-sampled as planes;
//   ||u_P||^2 = sum_{i,j} w_i w_j (g_i . g_j)  -> the 10 dot products of every 2x2 cell of g, 5 records per pixel (this kernel).
// g: padded NHWC fp16 [B, H+2, W+2, C] (border = finite values, weight 0).  gram [B, H, W, 5] fp32 per pixel q = (y, x):
//   {g_q.g_q, g_q.g_(y,x+1), g_q.g_(y+1,x), g_q.g_(y+1,x+1), g_(y,x+1).g_(y+1,x)}.  One wave per run of 4 pixels of a row.
template <int NV>
__global__ __launch_bounds__(256) void pixel_gram_kernel(const uint16_t* __restrict__ g, float* __restrict__ gram, int B, int H, int W, int C) {
    const int lane = threadIdx.x & 63;
    const int w4 = (W + 3) / 4;
    const size_t nrun = (size_t)B * H * w4;
    const size_t rowp = (size_t)(W + 2) * C;
    for (size_t run = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); run < nrun; run += (size_t)gridDim.x * 4) {
        const int x0 = (int)(run % w4) * 4;
        const int y = (int)((run / w4) % H);
        const int b = (int)(run / ((size_t)w4 * H));
        const uint16_t* r0 = g + ((size_t)b * (H + 2) + y + 1) * rowp + (size_t)(x0 + 1) * C;
        const uint16_t* r1 = r0 + rowp;
        float acc[4][5];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 0; d < 5; ++d) acc[i][d] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (lane + 64 * v) * 8;
            if (c >= C) continue;
            float t[5][8], u[5][8];                       // rows y and y+1, columns x0 .. x0+4 (clamped to the padded row: x <= W)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int xj = x0 + j <= W ? j : W - x0;
                const uint4 a = *reinterpret_cast<const uint4*>(r0 + (size_t)xj * C + c), bq = *reinterpret_cast<const uint4*>(r1 + (size_t)xj * C + c);
                const uint16_t *ea = reinterpret_cast<const uint16_t*>(&a), *eb = reinterpret_cast<const uint16_t*>(&bq);
#pragma unroll
                for (int e = 0; e < 8; ++e) { t[j][e] = f16_to_f32(ea[e]); u[j][e] = f16_to_f32(eb[e]); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc[i][0] += t[i][e] * t[i][e];
                    acc[i][1] += t[i][e] * t[i + 1][e];
                    acc[i][2] += t[i][e] * u[i][e];
                    acc[i][3] += t[i][e] * u[i + 1][e];
                    acc[i][4] += t[i + 1][e] * u[i][e];
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                float s = acc[i][d];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                acc[i][d] = s;
            }
        if (lane < 20) {
            const int i = lane / 5, d = lane - 5 * i;
            float val = 0.f;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int dd = 0; dd < 5; ++dd) if (ii == i && dd == d) val = acc[ii][dd];
            if (x0 + i < W) gram[(((size_t)b * H + y) * W + x0 + i) * 5 + d] = val;
        }
    }
}
// scale[b, Y, X] = s / ||u_P||  on the (2H, 2W) grid, from the 2x2-cell dot products: ||u||^2 = sum_i w_i^2 (g_i.g_i) + 2 sum_{i<j} w_i w_j (g_i.g_j)
// `flag` (optional): set to 1 when a pixel's squared norm is not finite -- an inf / NaN anywhere in the 16-bit image tower (fp16 operands
// saturate at 65504) has reached the head feature map; the engine's always-on overflow sentinel (Engine::overflow_seen)
__global__ void norm_scale_plane_kernel(const float* __restrict__ gram, float* __restrict__ scale, int B, int H, int W, float s, unsigned* __restrict__ flag) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t n = (size_t)B * Ho * Wo;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), b = (int)(i / ((size_t)Wo * Ho));
        const float sy = ry * (float)yo, sx = rx * (float)xo;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float wa = (1.f - ly) * (1.f - lx), wb = (1.f - ly) * lx, wc = ly * (1.f - lx), wd = ly * lx;
        const float* ra = gram + (((size_t)b * H + y0) * W + x0) * 5;          // a = (y0, x0): right -> a.b, down -> a.c, diag -> a.d, anti -> b.c
        const float* rb = gram + (((size_t)b * H + y0) * W + x1) * 5;          // b = (y0, x1): down -> b.d
        const float* rc = gram + (((size_t)b * H + y1) * W + x0) * 5;          // c = (y1, x0): right -> c.d
        const float* rd = gram + (((size_t)b * H + y1) * W + x1) * 5;
        // a clamped tap (x1 == x0 or y1 == y0) carries weight exactly 0, so the records it would mis-address never count
        const float n2 = wa * wa * ra[0] + wb * wb * rb[0] + wc * wc * rc[0] + wd * wd * rd[0] +
                         2.f * (wa * wb * ra[1] + wa * wc * ra[2] + wa * wd * ra[3] + wb * wc * ra[4] + wb * wd * rb[2] + wc * wd * rc[1]);
        scale[i] = s * rsqrtf(n2);
        if (flag && !(fabsf(n2) <= 3.0e38f)) *flag = 1u;        // (benign race: every writer stores the same value)
    }
}
// x2 bilinear (align_corners=True) of the label planes R (padded [P, H+2, W+2] fp32, interior read) with the per-pixel factor and the
// reference's fp16 rounding of the logits: low[p, Y, X] = fp16( scale[p / K, Y, X] * bilinear(R[p])(Y, X) )  -> [P, 2H, 2W] fp32.
// A block = one plane x a band of LB output rows: the R rows under the band are staged in LDS, the band's pixels are spread over
// all threads (at W = 120 one output row is only 60 float4 stores).
constexpr int UPS_LB = 8;
__device__ __forceinline__ int ups_stage_r(const float* __restrict__ in, float* __restrict__ Rr, int pl, int H, int W, float ry, int ya, int yb) {
    const int r_lo = (int)(ry * (float)ya);
    int r_hi = (int)(ry * (float)yb) + 1;
    if (r_hi > H - 1) r_hi = H - 1;
    const int nR = r_hi - r_lo + 1;
    for (int i = threadIdx.x; i < nR * W; i += blockDim.x) {
        const int r = i / W, x = i - r * W;
        Rr[i] = in[((size_t)pl * (H + 2) + r_lo + r + 1) * (W + 2) + 1 + x];
    }
    return r_lo;
}
// Thread layout of the banded kernels: a row of `n` items takes n threads, 256 / n rows are in flight per pass (n <= 256), so a thread
// keeps ONE column: its source columns and weights are computed once, the row loop only adds the row terms.
struct ColTerms { int x0[4], x1[4]; float lx[4]; };
__device__ __forceinline__ ColTerms col_terms4(float rx, int x4, int Win) {
    ColTerms t;
#pragma unroll
    for (int e = 0; e < 4; ++e) src_tap(rx, x4 * 4 + e, Win, t.x0[e], t.x1[e], t.lx[e]);
    return t;
}
// four scaled low-resolution logits of row Y (the association of upsample_bilinear2d, then the fp16 rounding of `half @ half`)
__device__ __forceinline__ void ups_low4(const float* __restrict__ Rr, const float* __restrict__ sc, int H, int W, float ry, int r_lo, int Y,
                                         int x4, const ColTerms& t, float (&o)[4]) {
    int y0, y1;
    float ly;
    src_tap(ry, Y, H, y0, y1, ly);
    const float* q0 = Rr + (y0 - r_lo) * W;
    const float* q1 = Rr + (y1 - r_lo) * W;
    const float4 sv = *reinterpret_cast<const float4*>(sc + (size_t)Y * (2 * W) + x4 * 4);
    const float se[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = round_f16(se[e] * bilerp(q0[t.x0[e]], q0[t.x1[e]], q1[t.x0[e]], q1[t.x1[e]], t.lx[e], ly));
}
__global__ __launch_bounds__(256) void upsample2x_planes_scaled_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                                       float* __restrict__ out, int P, int K, int H, int W) {
    extern __shared__ float Rr[];             // [<= UPS_LB/2 + 4][W]
    const int Ho = 2 * H, Wo = 2 * W, w4 = Wo / 4, bands = (Ho + UPS_LB - 1) / UPS_LB;
    const int pl = blockIdx.x / bands, ya = (blockIdx.x - pl * bands) * UPS_LB;
    const int yb = ya + UPS_LB - 1 < Ho - 1 ? ya + UPS_LB - 1 : Ho - 1;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    const int r_lo = ups_stage_r(in, Rr, pl, H, W, ry, ya, yb);
    __syncthreads();
    const float* sc = scale + (size_t)(pl / K) * Ho * Wo;
    const int nsub = w4 < 256 ? 256 / w4 : 1;
    for (int x4 = threadIdx.x % (w4 < 256 ? w4 : 256); x4 < w4; x4 += 256) {
        const int sub = w4 < 256 ? threadIdx.x / w4 : 0;
        if (sub >= nsub) break;
        const ColTerms t = col_terms4(rx, x4, W);
        for (int Y = ya + sub; Y <= yb; Y += nsub) {
            float o[4];
            ups_low4(Rr, sc, H, W, ry, r_lo, Y, x4, t, o);
            *reinterpret_cast<float4*>(out + ((size_t)pl * Ho + Y) * Wo + x4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
// The same followed by scratch.output_conv's x2 bilinear (lseg_net.py:203) in one pass: R [P, H+2, W+2] -> logits [P, 4H, 4W], the
// (2H, 2W) logits only ever exist as a band in LDS.  Bit-identical to upsample2x_planes_scaled + upsample2x_planes (bilerp).
// A block = one plane x the output rows whose upper source row lies in a band of LB low rows: phase 1 stages the R rows under the band
// [Rr], phase 2 the band of (2H, 2W) logits, LB + 1 rows [Lr]: low = fp16(scale * bilerp(R)), phase 3 writes the output rows.
// Phase 3, ROLLING ROWS (round 4): a thread owns one group of 4 output columns and walks CONSECUTIVE output rows, keeping the horizontal
// interpolation h = fma(lx, b, (1 - lx) a) -- the first half of bilerp() -- of its two current source rows in registers; a new h row is
// formed only when the source row advances (every second output row): a quarter of the horizontal work and of the LDS reads of the
// direct form (bilerp() of four Lr taps per output pixel: 16 ds_read_b32 + ~65 VALU instructions per 16 bytes stored), no further LDS
// image, same occupancy, same operations in the same association = same bits (test_one_pass_x4_upsample_equals_its_two_stages).
// tools/upsample_bench.py, B = 36, one box (profiles/r04_head_kernels.txt): direct 1372 / 1258 us, rolling 1184 / 1168 us (4.5 TB/s;
// B = 4: 115-118 -> 96-102 us); plain instead of non-temporal stores 1192 (B = 4: 136); rolling on a band of 8 rows 1268; a third LDS
// image of all h rows + aligned ds_read_b128 (49 KB of LDS: 3 blocks per CU instead of 7) 1354-1466.  A fill of the same 4.98 GB runs at
// 6.9 TB/s (tools/fill_bench.py): the kernel is still bound by its phase structure (three barriers, global -> LDS -> LDS -> HBM), not by
// the write.
constexpr int UPS4_LB = 16;
__device__ __forceinline__ float hlerp(float a, float b, float l) { return __builtin_fmaf(l, b, (1.f - l) * a); }   // bilerp()'s h0 / h1
template <int LB>
__global__ __launch_bounds__(256) void upsample4x_planes_scaled_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                                       float* __restrict__ out, int P, int K, int H, int W) {
    extern __shared__ float sm[];             // Rr [<= LB/2 + 4][W] | Lr [LB + 1][2W]
    const int Hl = 2 * H, Wl = 2 * W, Ho = 2 * Hl, Wo = 2 * Wl, w4 = Wo / 4, wl4 = Wl / 4, bands = (Hl + LB - 1) / LB;
    const int pl = blockIdx.x / bands, ya = (blockIdx.x - pl * bands) * LB;
    const int yb = ya + LB < Hl - 1 ? ya + LB : Hl - 1;        // last low row read (the band's rows + the next one)
    float* Rr = sm;
    float* Lr = sm + (LB / 2 + 4) * W;
    const float ry1 = (float)(H - 1) / (float)(Hl - 1), rx1 = (float)(W - 1) / (float)(Wl - 1);
    const int r_lo = ups_stage_r(in, Rr, pl, H, W, ry1, ya, yb);
    __syncthreads();
    const float* sc = scale + (size_t)(pl / K) * Hl * Wl;
    {   // the band of (2H, 2W) logits into LDS
        const int nsub = wl4 < 256 ? 256 / wl4 : 1;
        for (int x4 = threadIdx.x % (wl4 < 256 ? wl4 : 256); x4 < wl4; x4 += 256) {
            const int sub = wl4 < 256 ? threadIdx.x / wl4 : 0;
            if (sub >= nsub) break;
            const ColTerms t = col_terms4(rx1, x4, W);
            for (int Y = ya + sub; Y <= yb; Y += nsub) {
                float o[4];
                ups_low4(Rr, sc, H, W, ry1, r_lo, Y, x4, t, o);
                *reinterpret_cast<float4*>(Lr + (Y - ya) * Wl + x4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    __syncthreads();
    const float ry = (float)(Hl - 1) / (float)(Ho - 1), rx = (float)(Wl - 1) / (float)(Wo - 1);
    // output rows of this band: y0(yo) = floor(ry * yo) in [ya, ya + LB)
    int yo_first = (int)ceilf((float)ya / ry);
    while (yo_first > 0 && (int)(ry * (float)(yo_first - 1)) >= ya) --yo_first;
    while ((int)(ry * (float)yo_first) < ya) ++yo_first;
    int yo_end = yo_first;
    while (yo_end < Ho && (int)(ry * (float)yo_end) < ya + LB) ++yo_end;
    const int nsub = w4 < 256 ? 256 / w4 : 1;
    const int per = (yo_end - yo_first + nsub - 1) / nsub;       // sub-group `sub` takes a contiguous share of the band's output rows
    for (int x4 = threadIdx.x % (w4 < 256 ? w4 : 256); x4 < w4; x4 += 256) {
        const int sub = w4 < 256 ? threadIdx.x / w4 : 0;
        if (sub >= nsub) break;
        const ColTerms t = col_terms4(rx, x4, Wl);
        const int ys = yo_first + sub * per, ye = ys + per < yo_end ? ys + per : yo_end;
        float h0[4] = {0.f, 0.f, 0.f, 0.f}, h1[4] = {0.f, 0.f, 0.f, 0.f};
        int c0 = -1, c1 = -1;                                // the source rows h0 / h1 hold
        auto hrow = [&](int y, float (&h)[4]) {
            const float* q = Lr + (y - ya) * Wl;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = hlerp(q[t.x0[e]], q[t.x1[e]], t.lx[e]);
        };
        for (int yo = ys; yo < ye; ++yo) {
            int y0, y1;
            float ly;
            src_tap(ry, yo, Hl, y0, y1, ly);
            if (y0 != c0) {
                if (y0 == c1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h0[e] = h1[e];
                } else {
                    hrow(y0, h0);
                }
                c0 = y0;
            }
            if (y1 != c1) {
                if (y1 == c0) {                                // the clamped last row: both taps on one source row
#pragma unroll
                    for (int e = 0; e < 4; ++e) h1[e] = h0[e];
                } else {
                    hrow(y1, h1);
                }
                c1 = y1;
            }
            const float wy = 1.f - ly;
            const f32x4_t ov = {__builtin_fmaf(ly, h1[0], wy * h0[0]), __builtin_fmaf(ly, h1[1], wy * h0[1]),
                                __builtin_fmaf(ly, h1[2], wy * h0[2]), __builtin_fmaf(ly, h1[3], wy * h0[3])};
            __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t*>(out + ((size_t)pl * Ho + yo) * Wo) + x4);     // written once, never re-read by the engine
        }
    }
}

// ---- text: token+positional embedding in fp16 steps ([3P] clip/model.py encode_text) ---------------
__global__ void text_embed_kernel(const int64_t* tok, const float* emb, const float* pos, uint16_t* x,
                                  int rows, int L, int ctx, int W) {
    // rows = K * L: only the first L <= ctx positions of every label are embedded (tokens are [K, ctx])
    const unsigned total = (unsigned)rows * W;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int d = (int)(idx % (unsigned)W);
        const int r = (int)(idx / (unsigned)W);
        const int k = r / L, l = r - k * L;
        const float e = round_f16(emb[(size_t)tok[(size_t)k * ctx + l] * W + d]);
        const float p = round_f16(pos[(size_t)l * W + d]);
        x[idx] = f32_to_f16(e + p);
    }
}
// pooled[k,:] = x[k*L + eot[k], :]
__global__ void text_pool_kernel(const uint16_t* x, const int* eot, uint16_t* pooled, int K, int L, int W) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * W) return;
    const int k = idx / W, d = idx - k * W;
    pooled[idx] = x[((size_t)k * L + eot[k]) * W + d];
}
// t / ||t|| on an fp16 tensor (lseg_net.py:192): norm accumulates in fp32, is rounded to fp16,
// the quotient is rounded to fp16.  One wave per row.
__global__ __launch_bounds__(256) void text_l2norm_kernel(const uint16_t* t, uint16_t* out, int K, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= K) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = f16_to_f32(t[(size_t)row * C + c]); s += v * v; }
    const float nrm = round_f16(sqrtf(wave_sum(s)));
    for (int c = lane; c < C; c += 64) out[(size_t)row * C + c] = f32_to_f16(f16_to_f32(t[(size_t)row * C + c]) / nrm);
}

// ---- one-off parameter repacking -----------------------------------------------------------------------
// generic strided convert: out[i] = (T) in[i] (any of f32/f16/bf16 -> f32/f16/bf16)
__global__ void convert_kernel(const void* in, int in_dtype, void* out, int out_dtype, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        store_from_f32(out, i, out_dtype, load_as_f32(in, i, in_dtype));
}
// Range scan of a 16-bit buffer (Engine::check_range): counts non-finite values and finite values with |x| >= 2^15, tracks the largest
// finite |x|.  16 bytes per lane; one atomic per workgroup and counter.
__global__ void range16_kernel(const uint16_t* p, size_t n, int dtype, unsigned long long* out) {
    unsigned long long bad = 0, near = 0;
    float mx = 0.f;
    const size_t n8 = n >> 3;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8 + (n & 7); i += (size_t)gridDim.x * blockDim.x) {
        uint16_t v[8];
        int cnt = 8;
        if (i < n8) {
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(p + i * 8);
        } else {
            cnt = 1;
            v[0] = p[n8 * 8 + (i - n8)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= cnt) break;
            const float f = dtype == DT_F16 ? f16_to_f32(v[j]) : bf16_to_f32(v[j]);
            const float a = fabsf(f);
            if (!(a <= 3.0e38f)) ++bad;                 // inf or NaN
            else {
                if (a >= 32768.f) ++near;
                mx = fmaxf(mx, a);
            }
        }
    }
    __shared__ unsigned long long sb[2];
    __shared__ unsigned int sm;
    if (threadIdx.x == 0) { sb[0] = sb[1] = 0; sm = 0; }
    __syncthreads();
    if (bad) atomicAdd(&sb[0], bad);
    if (near) atomicAdd(&sb[1], near);
    atomicMax(&sm, __float_as_uint(mx));                // non-negative floats order like their bit patterns
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sb[0]) atomicAdd(&out[0], sb[0]);
        if (sb[1]) atomicAdd(&out[1], sb[1]);
        atomicMax(&out[2], (unsigned long long)sm);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&out[3], (unsigned long long)n);
}
// 2-D convert with a padded output row stride: in [R, C] -> out [R, ld] (columns >= C untouched)
__global__ void convert2d_kernel(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, int ld) {
    const size_t n = (size_t)R * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (size_t)r * C);
        store_from_f32(out, (size_t)r * ld + c, out_dtype, load_as_f32(in, i, in_dtype));
    }
}
// transpose-convert: in [R, C] -> out [C, R]
__global__ void transpose_convert_kernel(const void* in, int in_dtype, void* out, int out_dtype, int R, int C) {
    const size_t n = (size_t)R * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / R), r = (int)(i - (size_t)c * R);     // out index i = c*R + r
        store_from_f32(out, i, out_dtype, load_as_f32(in, (size_t)r * C + c, in_dtype));
    }
}
// conv3x3 weight [Co,Ci,3,3] (+ eval BatchNorm fold) -> [Co, 9*Ci] tap-major; bias_out = beta - mean*s
// (ResidualConvUnit_custom conv+bn pairs, lseg_blocks.py:276-283)
__global__ void pack_conv3x3_kernel(const float* w, const float* bn_w, const float* bn_b, const float* bn_m,
                                    const float* bn_v, float bn_eps, const float* conv_bias, void* wp,
                                    float* bias_out, int Co, int Ci, int Cip, int dtype) {
    const size_t n = (size_t)Co * 9 * Ci;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const int tap = (int)((i / Ci) % 9);
        const int co = (int)(i / ((size_t)9 * Ci));
        const float s = bn_w ? bn_w[co] * rsqrtf(bn_v[co] + bn_eps) : 1.f;
        const float v = w[((size_t)co * Ci + ci) * 9 + tap] * s;
        store_from_f32(wp, ((size_t)co * 9 + tap) * Cip + ci, dtype, v);   // Cip >= Ci: zero-padded K
        if (ci == 0 && tap == 0 && bias_out) {
            float b = conv_bias ? conv_bias[co] * s : 0.f;
            if (bn_w) b += bn_b[co] - bn_m[co] * s;
            bias_out[co] = b;
        }
    }
}
// ConvTranspose2d(k = s) weight [Ci, Co, s, s] -> GEMM weight [(i*s + j)*Co + co, Ci]
__global__ void pack_convT_kernel(const float* w, void* wp, int Ci, int Co, int Cp, int s, int dtype) {
    const size_t n = (size_t)s * s * Co * Ci;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Ci);
        const size_t row = idx / Ci;
        const int co = (int)(row % Co);
        const int ij = (int)(row / Co);
        const int i = ij / s, j = ij - i * s;
        const float v = w[(((size_t)ci * Co + co) * s + i) * s + j];
        store_from_f32(wp, ((size_t)ij * Cp + co) * Cp + ci, dtype, v);   // [(ij*Cp+co), Cp], zero-padded
    }
}

// ---- test taps: NHWC 16-bit (optionally padded) -> NCHW fp32 -------------------------------------------------
__global__ void nhwc_to_nchw_f32_kernel(const uint16_t* in, float* out, int B, int H, int W, int C, int Cs, int pad, int dtype, size_t lo_plane) {
    const size_t n = (size_t)B * C * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const int b = (int)(i / ((size_t)W * H * C));
        const size_t src = (((size_t)b * (H + 2 * pad) + y + pad) * (W + 2 * pad) + x + pad) * Cs + c;   // Cs = channel stride
        out[i] = load_as_f32(in, src, dtype) + (lo_plane ? load_as_f32(in, src + lo_plane, dtype) : 0.f);
    }
}
// fp32 [M, C] rows (pixel-major) -> NCHW fp32
__global__ void rows_to_nchw_f32_kernel(const float* in, float* out, int B, int HW, int C) {
    const size_t n = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const int b = (int)(i / ((size_t)HW * C));
        out[i] = in[((size_t)b * HW + p) * C + c];
    }
}

// ---- arch_option 1/2 head blocks on label planes (lseg_net.py:29-79) ----------------------------------------
// out[b,k,y,x] = act( conv3x3_1ch(in[b,k]) (+ max_k in[b,:,y,x] for the bottleneck) )
__global__ void head_block_kernel(const float* in, float* out, const float* w9, const float* bias, int B, int K,
                                  int H, int W, int bottleneck, int act, int apply_act) {
    const size_t n = (size_t)B * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int b = (int)(i / ((size_t)W * H));
        float mx = -INFINITY;
        if (bottleneck)
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, in[(((size_t)b * K + k) * H + y) * W + x]);
        for (int k = 0; k < K; ++k) {
            const float* pl = in + ((size_t)b * K + k) * H * W;
            float acc = bias[0];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += w9[(dy + 1) * 3 + dx + 1] * pl[(size_t)yy * W + xx];
                }
            if (bottleneck) acc += mx;
            if (apply_act) {
                if (act == 0) acc = fmaxf(acc, 0.f);
                else if (act == 1) acc = acc > 0.f ? acc : 0.01f * acc;
                else acc = tanhf(acc);
            }
            out[(((size_t)b * K + k) * H + y) * W + x] = acc;
        }
    }
}

// ---- argmax over K label planes: in fp32 [B,K,HW] -> uint8 [B,HW] (first max wins, like torch.max) ---------
__global__ void argmax_planes_kernel(const float* in, uint8_t* out, int B, int K, int HW) {
    const size_t n = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int b = (int)(i / HW);
        float best = -INFINITY; int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float v = in[((size_t)b * K + k) * HW + p];
            if (v > best) { best = v; bi = k; }
        }
        out[i] = (uint8_t)bi;
    }
}

// LayerNorm backward (timm norm1/norm2, fp32 input x): one wave per row, rows strided over the grid so that a wave
// keeps its dgamma / dbeta partial sums in registers and issues one atomic per column at the end.
//   xh = (x - mean) * rstd ; g = dy * gamma ; dx = rstd * (g - mean(g) - xh * mean(g * xh)) ; dgamma += dy * xh ; dbeta += dy
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy, int dy_dtype, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int M, int D, float eps, int accumulate, float* __restrict__ partial,
                                                            uint16_t* __restrict__ dx16) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    float4 ga[MAXV], dg[MAXV], db[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        ga[i] = g < nv ? reinterpret_cast<const float4*>(gamma)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = wave; row < M; row += nwaves) {
        float4 xv[MAXV], gv[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int g = lane + 64 * i;
            if (g < nv) {
                xv[i] = reinterpret_cast<const float4*>(x + (size_t)row * D)[g];
                if (dy_dtype == DT_F32) {
                    gv[i] = reinterpret_cast<const float4*>((const float*)dy + (size_t)row * D)[g];
                } else {
                    const uint2 u = reinterpret_cast<const uint2*>((const uint16_t*)dy + (size_t)row * D)[g];
                    gv[i].x = load_as_f32(&u, 0, dy_dtype); gv[i].y = load_as_f32(&u, 1, dy_dtype);
                    gv[i].z = load_as_f32(&u, 2, dy_dtype); gv[i].w = load_as_f32(&u, 3, dy_dtype);
                }
                s += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
            } else {
                xv[i] = make_float4(0.f, 0.f, 0.f, 0.f); gv[i] = xv[i];
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + 64 * i < nv) {
                const float a = xv[i].x - mean, b = xv[i].y - mean, c = xv[i].z - mean, d = xv[i].w - mean;
                q += a * a + b * b + c * c + d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + 64 * i < nv) {
                // xv <- xhat ; gv stays dy ; accumulate the parameter gradients ; t = dy * gamma
                xv[i].x = (xv[i].x - mean) * rstd; xv[i].y = (xv[i].y - mean) * rstd;
                xv[i].z = (xv[i].z - mean) * rstd; xv[i].w = (xv[i].w - mean) * rstd;
                dg[i].x += gv[i].x * xv[i].x; dg[i].y += gv[i].y * xv[i].y; dg[i].z += gv[i].z * xv[i].z; dg[i].w += gv[i].w * xv[i].w;
                db[i].x += gv[i].x; db[i].y += gv[i].y; db[i].z += gv[i].z; db[i].w += gv[i].w;
                gv[i].x *= ga[i].x; gv[i].y *= ga[i].y; gv[i].z *= ga[i].z; gv[i].w *= ga[i].w;
                sg += gv[i].x + gv[i].y + gv[i].z + gv[i].w;
                sgx += gv[i].x * xv[i].x + gv[i].y * xv[i].y + gv[i].z * xv[i].z + gv[i].w * xv[i].w;
            }
        }
        const float a = wave_sum(sg) / (float)D, b = wave_sum(sgx) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int g = lane + 64 * i;
            if (g < nv) {
                float4* o = reinterpret_cast<float4*>(dx + (size_t)row * D) + g;
                float4 r = make_float4(rstd * (gv[i].x - a - xv[i].x * b), rstd * (gv[i].y - a - xv[i].y * b),
                                       rstd * (gv[i].z - a - xv[i].z * b), rstd * (gv[i].w - a - xv[i].w * b));
                if (accumulate) { const float4 p = *o; r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w; }
                *o = r;
                if (dx16) {          // the 16-bit copy the next Linear backward reads as its dY (saves a conversion pass)
                    uint2 pk;
                    pk.x = pack2_dt(r.x, r.y, dy_dtype); pk.y = pack2_dt(r.z, r.w, dy_dtype);
                    reinterpret_cast<uint2*>(dx16 + (size_t)row * D)[g] = pk;
                }
            }
        }
    }
    // parameter gradients: with a workspace, the four waves of the block are summed through LDS and every block writes ONE
    // partial row [2D] (reduced by colreduce_kernel: deterministic, no atomics); without, fp32 atomics (a few thousand waves
    // hammering 2D addresses: 0.5 ms at M = 7208, D = 1024 -- 25x the streaming time)
    if (partial) {
        __shared__ float red[4][2][64 * 4 * MAXV > 1024 ? 1024 : 64 * 4 * MAXV];
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int g = lane + 64 * i;
            if (g < nv) {
                reinterpret_cast<float4*>(red[w][0])[g] = dg[i];
                reinterpret_cast<float4*>(red[w][1])[g] = db[i];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            partial[(size_t)blockIdx.x * 2 * D + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
            partial[(size_t)blockIdx.x * 2 * D + D + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            atomicAdd(dgamma + 4 * g + 0, dg[i].x); atomicAdd(dgamma + 4 * g + 1, dg[i].y);
            atomicAdd(dgamma + 4 * g + 2, dg[i].z); atomicAdd(dgamma + 4 * g + 3, dg[i].w);
            atomicAdd(dbeta + 4 * g + 0, db[i].x); atomicAdd(dbeta + 4 * g + 1, db[i].y);
            atomicAdd(dbeta + 4 * g + 2, db[i].z); atomicAdd(dbeta + 4 * g + 3, db[i].w);
        }
    }
}

// ---- building blocks of the backward pass (first bricks of SURVEY.md §8 a17) -----------------------------------------
// out[c, r] = in[r, c] for r < R (zero for R <= r < ldo): 16-bit matrices, 64x64 tiles through LDS.  Used to put the
// contraction dimension of dgrad / wgrad GEMMs on the fast axis (dY^T, X^T with M padded to the GEMM's K-step).
// `shift`: out[c, r] = in[r + shift, c] (zero outside [0, R)) -- the 9 taps of a 3x3 conv's wgrad are row shifts of the
// padded NHWC activation.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                         int R, int C, int ldi, int ldo, int shift, int relu) {
    __shared__ uint16_t t[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63, r = r0 + rr + shift, c = c0 + cc;
        uint16_t v = (r >= 0 && r < R && c < C) ? in[(size_t)r * ldi + c] : (uint16_t)0;
        if (relu && (v & 0x8000)) v = 0;          // relu != 0: the operand is ReLU(in) (the RCU convs read their input through a ReLU)
        t[rr][cc] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, rr = i & 63, r = r0 + rr, c = c0 + cc;
        if (c < C && r < ldo) out[(size_t)c * ldo + r] = t[rr][cc];
    }
}
// the same with 16-byte accesses on both sides (C, ldi, ldo multiples of 8): a thread loads 8 consecutive columns of a row and stores 8
// consecutive rows of a column; the 2-byte shuffling happens in LDS
__device__ __forceinline__ void transpose16_vec_tile(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C, int ldi, int ldo,
                                                     int shift, int relu, int r0, int c0, uint16_t (*t)[72]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = threadIdx.x + k * 256;
        const int rr = i >> 3, c8 = (i & 7) * 8, r = r0 + rr + shift, c = c0 + c8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r >= 0 && r < R && c < C) v = *reinterpret_cast<const uint4*>(in + (size_t)r * ldi + c);
        if (relu) {
            uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = ((w[e] & 0x8000u) ? 0u : (w[e] & 0xffffu)) | ((w[e] & 0x80000000u) ? 0u : (w[e] & 0xffff0000u));
        }
        *reinterpret_cast<uint4*>(&t[rr][c8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = threadIdx.x + k * 256;
        const int cc = i >> 3, r8 = (i & 7) * 8, c = c0 + cc, r = r0 + r8;
        if (c < C && r < ldo) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (uint32_t)t[r8 + 2 * e][cc] | ((uint32_t)t[r8 + 2 * e + 1][cc] << 16);
            *reinterpret_cast<uint4*>(out + (size_t)c * ldo + r) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
__global__ __launch_bounds__(256) void transpose16_vec_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                             int R, int C, int ldi, int ldo, int shift, int relu) {
    __shared__ uint16_t t[64][72];
    transpose16_vec_tile(in, out, R, C, ldi, ldo, shift, relu, blockIdx.x * 64, blockIdx.y * 64, t);
}
// many transposes in one launch (the W^T copies of every Linear after an optimizer step): block -> (matrix, tile) through a
// table sorted by first block
__global__ __launch_bounds__(256) void transpose16_multi_kernel(const TransposeJob* __restrict__ jobs, int njobs) {
    __shared__ uint16_t t[64][72];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                     // last job whose first block <= blockIdx.x (uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TransposeJob j = jobs[lo];
    const int tile = (int)(blockIdx.x - j.blk0), tr = tile % j.tiles_r, tc = tile / j.tiles_r;
    transpose16_vec_tile(j.src, j.dst, j.R, j.C, j.C, j.R, 0, 0, tr * 64, tc * 64, t);
}
// 3x3 conv dgrad weights: Wd[ci, t', co] = Wp[co, 8 - t', ci]  (taps flipped, channels swapped) so that
// dX = conv3x3(dY, Wd) runs on the forward implicit-GEMM kernel.  Wp [Co, 9, Ci] tap-major (16-bit).
__global__ void conv_dgrad_pack_kernel(const uint16_t* __restrict__ wp, uint16_t* __restrict__ wd, int Co, int Ci) {
    const size_t n = (size_t)Co * 9 * Ci;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co), t = (int)((i / Co) % 9), ci = (int)(i / ((size_t)9 * Co));    // i indexes wd
        wd[i] = wp[((size_t)co * 9 + (8 - t)) * Ci + ci];
    }
}
// GELU (erf form, timm Mlp.act) backward: dx = dy * (Phi(z) + z * phi(z)), z = the pre-activation (fc1 output + bias).
// quick != 0: CLIP's QuickGELU x * sigmoid(1.702 x): d/dx = sg * (1 + 1.702 x (1 - sg)), sg = sigmoid(1.702 x)
__device__ __forceinline__ float gelu_grad(float z, int quick) {
    if (quick) {
        const float sg = 1.0f / (1.0f + __expf(-1.702f * z));
        return sg * (1.0f + 1.702f * z * (1.0f - sg));
    }
    return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}
__global__ void gelu_backward_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ pre, uint16_t* __restrict__ dx,
                                     size_t n, int dtype, int quick) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        store_from_f32(dx, i, dtype, load_as_f32(dy, i, dtype) * gelu_grad(load_as_f32(pre, i, dtype), quick));
}
// 16 bytes per lane on all three tensors (n % 8 == 0, 16-byte aligned bases)
__global__ void gelu_backward_vec_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ pre, uint16_t* __restrict__ dx,
                                         size_t n8, int dtype, int quick) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 ug = reinterpret_cast<const uint4*>(dy)[i], uz = reinterpret_cast<const uint4*>(pre)[i];
        const uint16_t *eg = reinterpret_cast<const uint16_t*>(&ug), *ez = reinterpret_cast<const uint16_t*>(&uz);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = load_as_f32(eg, k, dtype) * gelu_grad(load_as_f32(ez, k, dtype), quick);
        reinterpret_cast<uint4*>(dx)[i] = make_uint4(pack2_dt(v[0], v[1], dtype), pack2_dt(v[2], v[3], dtype),
                                                     pack2_dt(v[4], v[5], dtype), pack2_dt(v[6], v[7], dtype));
    }
}
// x2 bilinear (align_corners=True) backward, NHWC 16-bit: d_in (padded [B,H+2,W+2,C], interior written) gathers the
// tent-weighted d_out [B,2H,2W,C] -- the transpose of upsample2x_nhwc_kernel (refinenet upsample, lseg_blocks.py:352-354).
__global__ void upsample2x_nhwc_bwd_kernel(const uint16_t* __restrict__ dout, uint16_t* __restrict__ din, int B, int H, int W,
                                           int C, int dtype) {
    const int c8n = C / 8, Ho = 2 * H, Wo = 2 * W;
    const unsigned total = (unsigned)B * H * W * c8n;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % (unsigned)c8n);
        unsigned p = idx / (unsigned)c8n;
        const int x = (int)(p % (unsigned)W); p /= (unsigned)W;
        const int y = (int)(p % (unsigned)H);
        const int b = (int)(p / (unsigned)H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // output rows whose source coordinate lies within one pixel of y (forward: y0 = floor(sy), weights 1-ly / ly)
        const int ya = max(0, 2 * y - 2), yb = min(Ho - 1, 2 * y + 3), xa = max(0, 2 * x - 2), xb = min(Wo - 1, 2 * x + 3);
        for (int yo = ya; yo <= yb; ++yo) {
            const float sy = ry * (float)yo;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1);
            const float ly = sy - (float)y0;
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xa; xo <= xb; ++xo) {
                const float sx = rx * (float)xo;
                const int x0 = (int)sx, x1 = x0 + (x0 < W - 1);
                const float lx = sx - (float)x0;
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx == 0.f) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(dout + (((size_t)b * Ho + yo) * Wo + xo) * C + (size_t)c8 * 8);
                const uint16_t* e = reinterpret_cast<const uint16_t*>(&v);
                const float wgt = wy * wx;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += wgt * load_as_f32(e, k, dtype);
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2_dt(acc[2 * k], acc[2 * k + 1], dtype);
        *reinterpret_cast<uint4*>(din + (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + (size_t)c8 * 8) =
            make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// d logits of mean cross-entropy (CrossEntropyLoss(ignore_index) over [B,K,H,W]): (softmax_k - 1[k = t]) / n_valid, 0 at
// ignored pixels.  One thread per pixel, two passes over the K planes.
__global__ void softmax_ce_backward_kernel(const float* __restrict__ scores, const long long* __restrict__ target, float* __restrict__ dz,
                                           int K, int HW, size_t npix, int ignore_index, const double* __restrict__ nll) {
    const float inv_n = nll[1] > 0.0 ? (float)(1.0 / nll[1]) : 0.f;          // nll[1] = number of valid pixels (seg_stats)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const size_t b = i / HW;
        const float* col = scores + b * (size_t)K * HW + p;
        float* out = dz + b * (size_t)K * HW + p;
        const long long t = target[i];
        const bool valid = t != (long long)ignore_index && t >= 0 && t < K;
        if (!valid) {
            for (int k = 0; k < K; ++k) out[(size_t)k * HW] = 0.f;
            continue;
        }
        float m = -INFINITY, ssum = 0.f;
        for (int k = 0; k < K; ++k) {
            const float v = col[(size_t)k * HW];
            if (v > m) { ssum = ssum * __expf(m - v) + 1.f; m = v; } else { ssum += __expf(v - m); }
        }
        const float inv_s = 1.f / ssum;
        for (int k = 0; k < K; ++k)
            out[(size_t)k * HW] = (__expf(col[(size_t)k * HW] - m) * inv_s - (k == t ? 1.f : 0.f)) * inv_n;
    }
}
// ---- train-mode BatchNorm2d on the padded-NHWC maps (DPT ResidualConvUnit bn1/bn2, lseg_blocks.py:276-283, train()) ----
// The maps carry a zero border, so per-channel sums over ALL padded positions equal the sums over the image; n = B*H*W.
// Column statistics of 16-bit row-major matrices, 16 bytes per lane: a block = 32 column groups of 8 columns x 8 row lanes over
// `rows_per_block` rows; the row lanes are summed through LDS, then one fp32 atomic per column and block (out zeroed by the launcher) --
// or, with `partial` (deterministic reductions, lseg_config.flags bit 3), one partial row [gridDim.y][ncol] per row block, summed in a
// fixed order by colreduce_kernel.
//   MODE 0: out[c] += sum_r a[r,c]                                   (bias gradients)
//   MODE 1: out[c] += sum a ; out[C+c] += sum a^2                    (BatchNorm batch statistics)
//   MODE 2: out[c] += sum a ; out[C+c] += sum a * (b - mean_c) * rstd_c   (BatchNorm backward: a = dy, b = x, mean/rstd from `stats`)
template <int MODE>
__global__ __launch_bounds__(256) void colstats16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                         const float* __restrict__ stats, float inv_n, float eps, int dtype,
                                                         float* __restrict__ out, int R, int C, int ld, int rows_per_block,
                                                         float* __restrict__ partial) {
    __shared__ float red[MODE == 0 ? 1 : 2][8][32][8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + cg) * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mean[8], rstd[8];
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k < C ? c0 + k : C - 1;
            mean[k] = stats[c] * inv_n;
            rstd[k] = rsqrtf(fmaxf(stats[C + c] * inv_n - mean[k] * mean[k], 0.f) + eps);
        }
    }
    if (c0 + 8 <= C) {
        for (int r = r0 + rl; r < r1; r += 8) {
            const uint4 u = *reinterpret_cast<const uint4*>(a + (size_t)r * ld + c0);
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&u);
            if (MODE == 2) {
                const uint4 ux = *reinterpret_cast<const uint4*>(b + (size_t)r * ld + c0);
                const uint16_t* ex = reinterpret_cast<const uint16_t*>(&ux);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float g = load_as_f32(e, k, dtype);
                    s[k] += g;
                    q[k] += g * (load_as_f32(ex, k, dtype) - mean[k]) * rstd[k];      // border rows: g = 0
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = load_as_f32(e, k, dtype);
                    s[k] += v;
                    if (MODE == 1) q[k] += v * v;
                }
            }
        }
    } else if (c0 < C) {
        for (int r = r0 + rl; r < r1; r += 8)
            for (int k = 0; k < 8 && c0 + k < C; ++k) {
                const float v = load_as_f32(a, (size_t)r * ld + c0 + k, dtype);
                s[k] += v;
                if (MODE == 1) q[k] += v * v;
                if (MODE == 2) q[k] += v * (load_as_f32(b, (size_t)r * ld + c0 + k, dtype) - mean[k]) * rstd[k];
            }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][rl][cg][k] = s[k];
        if (MODE != 0) red[MODE == 0 ? 0 : 1][rl][cg][k] = q[k];
    }
    __syncthreads();
    const int k = threadIdx.x & 7, g = threadIdx.x >> 3;        // 256 threads = 32 groups x 8 columns
    const int c = (blockIdx.x * 32 + g) * 8 + k;
    if (c < C) {
        float t = 0.f, t2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            t += red[0][j][g][k];
            if (MODE != 0) t2 += red[MODE == 0 ? 0 : 1][j][g][k];
        }
        if (partial) {
            float* prow = partial + (size_t)blockIdx.y * (MODE == 0 ? C : 2 * C);
            prow[c] = t;
            if (MODE != 0) prow[C + c] = t2;
        } else {
            atomicAdd(&out[c], t);
            if (MODE != 0) atomicAdd(&out[C + c], t2);
        }
    }
}
__global__ void colsum16_scalar_kernel(const uint16_t* __restrict__ in, int dtype, float* __restrict__ out, int R, int C, int ld, int rows_per_block,
                                       float* __restrict__ partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += load_as_f32(in, (size_t)r * ld + c, dtype);
    if (partial) partial[(size_t)blockIdx.y * C + c] = s;
    else atomicAdd(&out[c], s);
}
// 0xffff per 16-bit half of a packed pair that is > 0 (bf16 / fp16 alike: sign bit clear, not zero)
__device__ __forceinline__ uint32_t relu_mask2(uint32_t x2) {
    return (((x2 & 0xffffu) != 0 && !(x2 & 0x8000u)) ? 0xffffu : 0u) | (((x2 >> 16) != 0 && !(x2 & 0x80000000u)) ? 0xffff0000u : 0u);
}
// y = gamma * (x - mean) * rstd + beta on the interior pixels (the border stays zero); mean/var from the batch sums (biased var)
// res1 / res2 (same geometry, may be NULL): y += res1 + res2 -- the RCU skip connection and the fusion add (lseg_blocks.py:288,347).
// inv_n = 1 / (pixels the sums were taken over): B*H*W, times the world size once the sums are all-reduced (SyncBatchNorm).
// 8 channels (16 bytes) per lane; C % 8 == 0.
__global__ void bn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int B, int H, int W, int C,
                                float eps, int dtype, float inv_n, const uint16_t* __restrict__ res1, const uint16_t* __restrict__ res2,
                                uint16_t* __restrict__ y_relu) {
    const int c8n = C >> 3;
    const size_t n = (size_t)B * H * W * c8n;
    // the grid stride is a multiple of C/8 for the power-of-two widths of the DPT head: a thread keeps ONE group of 8 channels, whose
    // mean / rstd / affine parameters are loaded (two float4 per array) and computed once, outside the pixel loop
    const bool fixed_c = ((size_t)gridDim.x * blockDim.x) % c8n == 0;
    float mean[8], rs[8], ga[8], be[8];
    auto load_c = [&](int c0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 s1 = *reinterpret_cast<const float4*>(stats + c0 + 4 * h), s2 = *reinterpret_cast<const float4*>(stats + C + c0 + 4 * h);
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + c0 + 4 * h), b4 = *reinterpret_cast<const float4*>(beta + c0 + 4 * h);
            const float a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = a1[k] * inv_n;
                mean[4 * h + k] = m;
                rs[4 * h + k] = rsqrtf(fmaxf(a2[k] * inv_n - m * m, 0.f) + eps);
                ga[4 * h + k] = gg[k]; be[4 * h + k] = bb[k];
            }
        }
    };
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (fixed_c && i < n) load_c((int)(i % c8n) * 8);
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % c8n) * 8;
        if (!fixed_c) load_c(c0);
        size_t p = i / c8n;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const int b = (int)(p / H);
        const size_t off = (((size_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C + c0;
        const uint4 ux = *reinterpret_cast<const uint4*>(x + off);
        uint4 u1 = make_uint4(0u, 0u, 0u, 0u), u2 = u1;
        if (res1) u1 = *reinterpret_cast<const uint4*>(res1 + off);
        if (res2) u2 = *reinterpret_cast<const uint4*>(res2 + off);
        const uint16_t *ex = reinterpret_cast<const uint16_t*>(&ux), *e1 = reinterpret_cast<const uint16_t*>(&u1), *e2 = reinterpret_cast<const uint16_t*>(&u2);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = ga[k] * (load_as_f32(ex, k, dtype) - mean[k]) * rs[k] + be[k];
            if (res1) v[k] += load_as_f32(e1, k, dtype);
            if (res2) v[k] += load_as_f32(e2, k, dtype);
        }
        const uint4 o = make_uint4(pack2_dt(v[0], v[1], dtype), pack2_dt(v[2], v[3], dtype), pack2_dt(v[4], v[5], dtype), pack2_dt(v[6], v[7], dtype));
        if (y) *reinterpret_cast<uint4*>(y + off) = o;
        if (y_relu)      // ReLU(y) for the 3x3 conv that reads it (and for that conv's weight gradient): negative halves cleared
            *reinterpret_cast<uint4*>(y_relu + off) = make_uint4(o.x & relu_mask2(o.x), o.y & relu_mask2(o.y), o.z & relu_mask2(o.z), o.w & relu_mask2(o.w));
    }
}
// dx = gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)) on the interior; bstats = [sum dy, sum dy * xhat]
__global__ void bn_bwd_apply_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, uint16_t* __restrict__ dx,
                                    const float* __restrict__ stats, const float* __restrict__ bstats, const float* __restrict__ gamma,
                                    int B, int H, int W, int C, float eps, int dtype, float inv_n) {
    const int c8n = C >> 3;
    const size_t n = (size_t)B * H * W * c8n;
    const bool fixed_c = ((size_t)gridDim.x * blockDim.x) % c8n == 0;      // see bn_apply_kernel
    float mean[8], rs[8], gr[8], mg[8], mq[8];
    auto load_c = [&](int c0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 s1 = *reinterpret_cast<const float4*>(stats + c0 + 4 * h), s2 = *reinterpret_cast<const float4*>(stats + C + c0 + 4 * h);
            const float4 t1 = *reinterpret_cast<const float4*>(bstats + c0 + 4 * h), t2 = *reinterpret_cast<const float4*>(bstats + C + c0 + 4 * h);
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + c0 + 4 * h);
            const float a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w}, b1[4] = {t1.x, t1.y, t1.z, t1.w},
                        b2[4] = {t2.x, t2.y, t2.z, t2.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = a1[k] * inv_n;
                const float r = rsqrtf(fmaxf(a2[k] * inv_n - m * m, 0.f) + eps);
                mean[4 * h + k] = m; rs[4 * h + k] = r; gr[4 * h + k] = gg[k] * r;
                mg[4 * h + k] = b1[k] * inv_n; mq[4 * h + k] = b2[k] * inv_n;
            }
        }
    };
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (fixed_c && i < n) load_c((int)(i % c8n) * 8);
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % c8n) * 8;
        if (!fixed_c) load_c(c0);
        size_t p = i / c8n;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const int b = (int)(p / H);
        const size_t off = (((size_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C + c0;
        const uint4 ux = *reinterpret_cast<const uint4*>(x + off), ug = *reinterpret_cast<const uint4*>(dy + off);
        const uint16_t *ex = reinterpret_cast<const uint16_t*>(&ux), *eg = reinterpret_cast<const uint16_t*>(&ug);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (load_as_f32(ex, k, dtype) - mean[k]) * rs[k];
            v[k] = gr[k] * (load_as_f32(eg, k, dtype) - mg[k] - xh * mq[k]);
        }
        *reinterpret_cast<uint4*>(dx + off) = make_uint4(pack2_dt(v[0], v[1], dtype), pack2_dt(v[2], v[3], dtype),
                                                         pack2_dt(v[4], v[5], dtype), pack2_dt(v[6], v[7], dtype));
    }
}
// ReLU backward: dx = dy where x > 0 (16-bit tensors of any shape); positive <=> sign bit clear and not zero (bf16 and fp16)
__global__ void relu_backward_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, uint16_t* __restrict__ dx, size_t n) {
    const size_t n8 = n >> 3;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 g = reinterpret_cast<const uint4*>(dy)[i], v = reinterpret_cast<const uint4*>(x)[i];
        reinterpret_cast<uint4*>(dx)[i] = make_uint4(g.x & relu_mask2(v.x), g.y & relu_mask2(v.y), g.z & relu_mask2(v.z), g.w & relu_mask2(v.w));
    }
    for (size_t i = (n8 << 3) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint16_t v = x[i];
        dx[i] = (v != 0 && !(v & 0x8000)) ? dy[i] : (uint16_t)0;
    }
}
// dx = (x > 0 ? dy : 0) + add : ReLU backward merged with the skip connection's gradient (RCU: out = f(relu(x)) + x)
__global__ void relu_backward_add_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, const uint16_t* __restrict__ add,
                                         uint16_t* __restrict__ dx, size_t n, int dtype) {
    const size_t n8 = n >> 3;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 ug = reinterpret_cast<const uint4*>(dy)[i], uv = reinterpret_cast<const uint4*>(x)[i], ua = reinterpret_cast<const uint4*>(add)[i];
        const uint16_t *eg = reinterpret_cast<const uint16_t*>(&ug), *ev = reinterpret_cast<const uint16_t*>(&uv), *ea = reinterpret_cast<const uint16_t*>(&ua);
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = ((ev[k] != 0 && !(ev[k] & 0x8000)) ? load_as_f32(eg, k, dtype) : 0.f) + load_as_f32(ea, k, dtype);
        reinterpret_cast<uint4*>(dx)[i] = make_uint4(pack2_dt(r[0], r[1], dtype), pack2_dt(r[2], r[3], dtype),
                                                     pack2_dt(r[4], r[5], dtype), pack2_dt(r[6], r[7], dtype));
    }
    for (size_t i = (n8 << 3) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint16_t v = x[i];
        const float g = (v != 0 && !(v & 0x8000)) ? load_as_f32(dy, i, dtype) : 0.f;
        store_from_f32(dx, i, dtype, g + load_as_f32(add, i, dtype));
    }
}
// ---- head-side backward bricks (lseg_net.py:185-203 under autograd) ----------------------------------------------------
// x2 bilinear (align_corners=True) backward of the logit planes, written straight as the GEMM operand of the correlation
// backward: d_low_rows[(b*h*w + p), k] (16-bit, leading dimension ldk >= K; padding columns must be pre-zeroed) from
// d_out [B,K,2h,2w] fp32 -- the transpose of upsample2x_planes_kernel (output_conv) fused with the planes->rows re-layout.
__global__ void upsample2x_planes_bwd_rows_kernel(const float* __restrict__ dout, uint16_t* __restrict__ rows, int B, int K, int H, int W,
                                                  int ldk, int dtype) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t n = (size_t)B * K * H * W;
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t p = i / W;
        const int y = (int)(p % H); p /= H;
        const int k = (int)(p % K);
        const int b = (int)(p / K);
        const float* plane = dout + ((size_t)b * K + k) * Ho * Wo;
        float acc = 0.f;
        const int ya = max(0, 2 * y - 2), yb = min(Ho - 1, 2 * y + 3), xa = max(0, 2 * x - 2), xb = min(Wo - 1, 2 * x + 3);
        for (int yo = ya; yo <= yb; ++yo) {
            const float sy = ry * (float)yo;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1);
            const float ly = sy - (float)y0;
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xa; xo <= xb; ++xo) {
                const float sx = rx * (float)xo;
                const int x0 = (int)sx, x1 = x0 + (x0 < W - 1);
                const float lx = sx - (float)x0;
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx != 0.f) acc += wy * wx * plane[(size_t)yo * Wo + xo];
            }
        }
        store_from_f32(rows, ((size_t)b * H * W + (size_t)y * W + x) * ldk + k, dtype, acc);
    }
}
// Fused backward of  CrossEntropyLoss(ignore_index)( output_conv(low) )  (lsegmentation_module.py:72 on lseg_net.py:203): the rows
// d_low_rows[(b*h*w + p), k] of the correlation's dY straight from the LOW-resolution logits `low` [B,K,h,w], the target mask
// [B,2h,2w] and the per-pixel log-sum-exp seg_stats saved -- the [B,K,2h,2w] logits and their gradient (2 x 138 MB per image at
// K = 150, 480x480) are never materialised:
//   d_low[b,k,y,x] = sum over the full-resolution pixels P whose bilinear footprint touches (y,x) of
//                    w(P -> y,x) * (exp(z_k(P) - lse(P)) - [k = t(P)]) / n_valid,     z_k(P) = the x2 bilinear of low[b,k] at P.
// With align_corners=True and an exact x2 grid the rows touching y are Y in [2y-1, 2y+2] (checked on the host for the geometry at
// hand), and every tap of those rows lies in the 3x3 neighbourhood of (y,x): one lane = one low-resolution pixel, 9 loads per class.
__global__ __launch_bounds__(256) void upsample_ce_bwd_rows_kernel(const float* __restrict__ low, const long long* __restrict__ target,
                                                                   const float* __restrict__ lse, const double* __restrict__ nll,
                                                                   uint16_t* __restrict__ rows, int B, int K, int H, int W, int ldk,
                                                                   int ignore_index, int dtype, const float* __restrict__ gscale) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t npix = (size_t)B * H * W;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int b = (int)(i / ((size_t)W * H));
    // d(mean CE)/d(logit) carries 1 / (valid pixels); gscale = the incoming d(loss) of an autograd caller (loss.backward(): 1)
    const float inv_n = (nll[1] > 0.0 ? (float)(1.0 / nll[1]) : 0.f) * (gscale ? *gscale : 1.f);
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    // the 4 candidate rows / columns: weight onto (y, x), and where their own two taps sit inside the 3x3 neighbourhood
    float wy[4], ly[4], wx[4], lx[4];
    int jy[4], jx[4];                       // index of the first tap relative to y-1 / x-1 (0 or 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int yo = 2 * y - 1 + r;
        wy[r] = 0.f; ly[r] = 0.f; jy[r] = 0;
        if (yo >= 0 && yo < Ho) {
            const float sy = ry * (float)yo;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1);
            const float l = sy - (float)y0;
            wy[r] = (y0 == y ? 1.f - l : 0.f) + (y1 == y ? l : 0.f);
            ly[r] = l; jy[r] = y0 - (y - 1);
        }
        const int xo = 2 * x - 1 + r;
        wx[r] = 0.f; lx[r] = 0.f; jx[r] = 0;
        if (xo >= 0 && xo < Wo) {
            const float sx = rx * (float)xo;
            const int x0 = (int)sx, x1 = x0 + (x0 < W - 1);
            const float l = sx - (float)x0;
            wx[r] = (x0 == x ? 1.f - l : 0.f) + (x1 == x ? l : 0.f);
            lx[r] = l; jx[r] = x0 - (x - 1);
        }
    }
    // per footprint pixel: coefficient (0 outside the image / ignored / zero weight), its log-sum-exp and its label
    float coef[4][4], lz[4][4];
    int tl[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int yo = 2 * y - 1 + r, xo = 2 * x - 1 + c;
            coef[r][c] = 0.f; lz[r][c] = 1e30f; tl[r][c] = -1;      // inactive: exp(z - 1e30) = 0
            const float wgt = wy[r] * wx[c];
            if (wgt != 0.f) {
                const size_t pi = ((size_t)b * Ho + yo) * Wo + xo;
                const long long t = target[pi];
                if (t != (long long)ignore_index && t >= 0 && t < K) { coef[r][c] = wgt * inv_n; lz[r][c] = lse[pi]; tl[r][c] = (int)t; }
            }
        }
    // clamped 3x3 neighbourhood offsets (rows / columns outside the map are never selected with a non-zero weight)
    const int ya = y > 0 ? y - 1 : 0, yb = y < H - 1 ? y + 1 : H - 1, xa = x > 0 ? x - 1 : 0, xb = x < W - 1 ? x + 1 : W - 1;
    const int ro[3] = {ya * W, y * W, yb * W}, co[3] = {xa, x, xb};
    const float* plane = low + (size_t)b * K * H * W;
    uint16_t* orow = rows + i * (size_t)ldk;
    for (int k0 = 0; k0 < ldk; k0 += 8) {
        float out[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = k0 + kk;
            float acc = 0.f;
            if (k < K) {
                const float* pl = plane + (size_t)k * H * W;
                float L[3][3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) L[a][c] = pl[ro[a] + co[c]];
                // horizontal interpolation of the three low rows at the 4 footprint columns (upsample_bilinear2d's association)
                float Hc[3][4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float p0 = jx[c] ? L[a][1] : L[a][0], p1 = jx[c] ? L[a][2] : L[a][1];
                        Hc[a][c] = (1.f - lx[c]) * p0 + lx[c] * p1;
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float q0 = jy[r] ? Hc[1][c] : Hc[0][c], q1 = jy[r] ? Hc[2][c] : Hc[1][c];
                        const float z = (1.f - ly[r]) * q0 + ly[r] * q1;
                        acc += coef[r][c] * (__expf(z - lz[r][c]) - (tl[r][c] == k ? 1.f : 0.f));
                    }
            }
            out[kk] = acc;
        }
        *reinterpret_cast<uint4*>(orow + k0) = make_uint4(pack2_dt(out[0], out[1], dtype), pack2_dt(out[2], out[3], dtype),
                                                          pack2_dt(out[4], out[5], dtype), pack2_dt(out[6], out[7], dtype));
    }
}
// backward of a = scale * x / ||x||_2 (row-wise; the fp16 roundings of the forward are treated as identity):
//   dx = (scale / ||x||) * (da - xh * (xh . da)),  xh = x / ||x||      x fp32 [M,C], da 16-bit, dx 16-bit
template <int MAXV>
__global__ __launch_bounds__(256) void l2norm_scale_bwd_kernel(const uint16_t* __restrict__ da, int da_dtype, const float* __restrict__ x,
                                                               uint16_t* __restrict__ dx, int dx_dtype, int M, int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;
    float4 xv[MAXV], gv[MAXV];
    float s = 0.f, d = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            xv[i] = reinterpret_cast<const float4*>(x + (size_t)row * C)[g];
            const uint2 u = reinterpret_cast<const uint2*>(da + (size_t)row * C)[g];
            gv[i].x = load_as_f32(&u, 0, da_dtype); gv[i].y = load_as_f32(&u, 1, da_dtype);
            gv[i].z = load_as_f32(&u, 2, da_dtype); gv[i].w = load_as_f32(&u, 3, da_dtype);
            if (da_dtype == DT_F16) {      // the reference's d(image_features.half()) = fp16(logit_scale * dA), a HALF tensor (lseg_net.py:194 under autograd)
                gv[i].x = round_f16(scale * gv[i].x); gv[i].y = round_f16(scale * gv[i].y);
                gv[i].z = round_f16(scale * gv[i].z); gv[i].w = round_f16(scale * gv[i].w);
            }
            s += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
            d += xv[i].x * gv[i].x + xv[i].y * gv[i].y + xv[i].z * gv[i].z + xv[i].w * gv[i].w;
        }
    }
    const float n2 = wave_sum(s);
    const float post = da_dtype == DT_F16 ? 1.f : scale;            // fp16 da: the scale was applied (and rounded) at the load
    const float inv = rsqrtf(n2), proj = wave_sum(d) / n2;          // xh . da / ||x|| = (x . da) / ||x||^2 ... applied to x below
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = lane + 64 * i;
        if (g < nv) {
            const float r[4] = {post * inv * (gv[i].x - xv[i].x * proj), post * inv * (gv[i].y - xv[i].y * proj),
                                post * inv * (gv[i].z - xv[i].z * proj), post * inv * (gv[i].w - xv[i].w * proj)};
            uint2 pk;
            pk.x = pack2_dt(r[0], r[1], dx_dtype);
            pk.y = pack2_dt(r[2], r[3], dx_dtype);
            reinterpret_cast<uint2*>(dx + (size_t)row * C)[g] = pk;
        }
    }
}
// dst[c] (+)= sum over the nb partial rows of src [nb, ld] (fixed order: deterministic).  Block = CW columns (16 bytes per thread) x
// 1024 / (CW / 4) row lanes, tree-summed through LDS; columns >= C go to dst2[c - C] (LayerNorm backward: one launch reduces
// [d gamma | d beta]).  CW = 16 puts 2D / 16 = 128 workgroups on the 8 MB of LayerNorm partials (with 64-column blocks 32 workgroups
// read them at 0.4 TB/s: 20 us per LayerNorm, 1 ms per training step).  C, C2, ld multiples of 4.
template <int CW>
__global__ __launch_bounds__(1024) void colreduce_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dst2,
                                                         int nb, int C, int C2, int ld, int accumulate) {
    constexpr int TPR = CW / 4, RL = 1024 / TPR;
    __shared__ float4 red[RL][TPR];
    const int cl = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int c = blockIdx.x * CW + cl * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C + C2)
        for (int r = rl; r < nb; r += RL) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[rl][cl] = s;
    __syncthreads();
#pragma unroll
    for (int h = RL / 2; h >= 1; h >>= 1) {
        if (rl < h) {
            const float4 o = red[rl + h][cl];
            float4 m = red[rl][cl];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            red[rl][cl] = m;
        }
        __syncthreads();
    }
    if (rl == 0 && c < C + C2) {
        float4 t = red[0][cl];
        float4* o = reinterpret_cast<float4*>(c < C ? dst + c : dst2 + (c - C));
        if (accumulate) { const float4 p = *o; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
        *o = t;
    }
}
// the same reduction for destinations that are not 16-byte aligned (parameter gradients bound to odd offsets of a caller's buffer)
__global__ __launch_bounds__(1024) void colreduce_scalar_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dst2,
                                                                int nb, int C, int C2, int ld, int accumulate) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C + C2)
        for (int r = rl; r < nb; r += 16) s += src[(size_t)r * ld + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C + C2) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][cl];
        float* o = c < C ? dst + c : dst2 + (c - C);
        *o = accumulate ? *o + t : t;
    }
}
// split-K partial results [nsplit][n] -> dst[n] (+)= sum_s part[s][n]   (n % 4 == 0; 16 bytes per lane)
__global__ void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ dst, int nsplit, size_t n4, size_t stride4, int accumulate) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(part)[i];
        for (int s = 1; s < nsplit; ++s) {
            const float4 b = reinterpret_cast<const float4*>(part)[(size_t)s * stride4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (accumulate) { const float4 d = reinterpret_cast<float4*>(dst)[i]; a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w; }
        reinterpret_cast<float4*>(dst)[i] = a;
    }
}

// ---- segmentation statistics on device -------------------------------------------------------------------------------
// One pass over the [B,K,H,W] fp32 scores: per pixel the arg-max label, its log-sum-exp and the score at the target;
// accumulates what the host-side metric code of the reference computes from the full logits tensor:
//   pixAcc / IoU counts  -- [3P] encoding/utils/metrics.py batch_pix_accuracy / batch_intersection_union
//                           (call sites lsegmentation_module.py:49-50,59-60; test_lseg.py:385-388)
//   cross-entropy        -- [3P] encoding/nn/loss.py SegmentationLosses = nn.CrossEntropyLoss(ignore_index)
//                           (call site lsegmentation_module.py:72): sum of -log_softmax(scores)[target] and the pixel count
// counts: [0] correct, [1] labeled, [2..2+K) area_inter, [2+K..2+2K) area_pred, [2+2K..2+3K) area_lab ; nll: [0] sum, [1] count.
// Integer counts are exact (atomics on integers); the NLL sum is a double atomic (order-dependent in the last bits).
// `up` != 0: `scores` is the LOW-resolution logits [B,K,h,w] and every pixel of the [B,2h,2w] output grid reads them through
// output_conv's x2 bilinear (align_corners=True, lseg_net.py:203) on the fly -- the 138 MB / image full-resolution logits are never
// written when the caller only wants masks and / or metrics.  `target` may be NULL (masks only), `argmax_out` (uint8 [B,2h,2w] or
// [B,H,W]) may be NULL (metrics only).
__global__ __launch_bounds__(256) void seg_stats_kernel(const float* __restrict__ scores, const long long* __restrict__ target,
                                                        int K, int HW, size_t npix, int ignore_index,
                                                        unsigned long long* __restrict__ counts, double* __restrict__ nll,
                                                        uint8_t* __restrict__ argmax_out, int up, int h, int w, float* __restrict__ lse_out) {
    extern __shared__ unsigned int hist[];                // [3K] per-block class histograms + [2] pixel counts
    for (int i = threadIdx.x; i < 3 * K + 2; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    double loss = 0.0;
    unsigned int nvalid = 0;
    const int Wo = 2 * w;
    const float ry = up ? (float)(h - 1) / (float)(2 * h - 1) : 0.f, rx = up ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const size_t b = i / HW;
        const float* col;
        size_t kstride;
        int o01 = 0, o10 = 0, o11 = 0;
        float ly = 0.f, lx = 0.f;
        if (up) {
            const int yo = p / Wo, xo = p - yo * Wo;
            int y0, y1, x0, x1;
            src_tap(ry, yo, h, y0, y1, ly);
            src_tap(rx, xo, w, x0, x1, lx);
            kstride = (size_t)h * w;
            col = scores + b * (size_t)K * kstride + (size_t)y0 * w + x0;
            o01 = x1 - x0; o10 = (y1 - y0) * w; o11 = o10 + o01;
        } else {
            kstride = (size_t)HW;
            col = scores + b * (size_t)K * HW + p;
        }
        const long long t = target ? target[i] : -1;
        float m = -INFINITY, ssum = 0.f, at_t = 0.f;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const float* c = col + (size_t)k * kstride;
            // same association as upsample_bilinear2d (and upsample2x_planes_kernel): bit-identical to the materialised logits
            const float v = up ? bilerp(c[0], c[o01], c[o10], c[o11], lx, ly) : c[0];
            if (v > m) {                                  // first maximum wins (torch.max / argmax tie rule)
                ssum = ssum * __expf(m - v) + 1.f;        // online log-sum-exp (exp(-inf) = 0 on the first label)
                m = v; arg = k;
            } else {
                ssum += __expf(v - m);
            }
            if (k == t) at_t = v;
        }
        if (argmax_out) argmax_out[i] = (uint8_t)arg;
        if (lse_out) lse_out[i] = m + __logf(ssum);          // saved for the fused cross-entropy backward
        if (!target) continue;
        const long long t1 = t + 1;                        // metrics.py: target + 1, predict + 1
        const int pred1 = arg + 1;
        if (t1 > 0) {
            atomicAdd(&hist[3 * K + 1], 1u);               // labeled
            atomicAdd(&hist[K + pred1 - 1], 1u);           // area_pred: predict * (target > 0)
            if (t1 <= K) atomicAdd(&hist[2 * K + (int)t1 - 1], 1u);   // area_lab (np.histogram range (1, nclass))
            if (pred1 == t1) {
                atomicAdd(&hist[3 * K], 1u);               // correct
                atomicAdd(&hist[pred1 - 1], 1u);           // area_inter
            }
        }
        if (t != (long long)ignore_index && t >= 0 && t < K) {
            loss += (double)(m + __logf(ssum) - at_t);
            ++nvalid;
        }
    }
    if (!target) return;                                   // uniform: nothing was accumulated
    // block reduction of the loss, then one atomic per block / per non-empty histogram bin
    __shared__ double lred[256];
    __shared__ unsigned int nred[256];
    lred[threadIdx.x] = loss; nred[threadIdx.x] = nvalid;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { lred[threadIdx.x] += lred[threadIdx.x + s]; nred[threadIdx.x] += nred[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && nred[0]) {
        atomicAdd(&nll[0], lred[0]);
        atomicAdd(&nll[1], (double)nred[0]);
    }
    for (int i = threadIdx.x; i < 3 * K + 2; i += blockDim.x) {
        const unsigned int v = hist[i];
        if (!v) continue;
        const int dst = i < 3 * K ? 2 + i : i - 3 * K;    // [3K] -> correct, [3K+1] -> labeled
        atomicAdd(&counts[dst], (unsigned long long)v);
    }
}

// ---- glue of the engine-level training step (csrc/train.hip) ------------------------------------------
// GELU (erf form) forward on a 16-bit tensor: the train-mode forward keeps the pre-activation (fc1 / readout Linear output)
// for the backward, so the activation is its own pass there.  8 elements per lane.
__global__ void gelu_forward_kernel(const uint16_t* __restrict__ pre, uint16_t* __restrict__ out, size_t n8, int dtype) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 u = reinterpret_cast<const uint4*>(pre)[i];
        const uint16_t* e = reinterpret_cast<const uint16_t*>(&u);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float z = load_as_f32(e, k, dtype); v[k] = 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
        reinterpret_cast<uint4*>(out)[i] = make_uint4(pack2_dt(v[0], v[1], dtype), pack2_dt(v[2], v[3], dtype),
                                                      pack2_dt(v[4], v[5], dtype), pack2_dt(v[6], v[7], dtype));
    }
}
// padded NHWC [B,H+2,W+2,C] -> rows [B*H*W, C] (16-bit, 16 bytes per lane): the GEMM-operand view of a gradient map
__global__ void unpad_rows_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int B, int H, int W, int C) {
    const int c8n = C / 8;
    const size_t total = (size_t)B * H * W * c8n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % c8n);
        size_t p = idx / c8n;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        reinterpret_cast<uint4*>(out)[idx] =
            *reinterpret_cast<const uint4*>(in + (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + (size_t)c8 * 8);
    }
}
// stride-2 conv backward helper: dYd (padded [B,H+2,W+2,C], pre-zeroed) gets dY (padded [B,Ho+2,Wo+2,C]) at the even
// positions: dYd[2oy, 2ox] = dY[oy, ox].  dgrad / wgrad of the stride-2 conv are then the stride-1 ones on dYd.
__global__ void dilate2_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dyd, int B, int Ho, int Wo, int H, int W, int C) {
    const int c8n = C / 8;
    const size_t total = (size_t)B * Ho * Wo * c8n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % c8n);
        size_t p = idx / c8n;
        const int x = (int)(p % Wo); p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        *reinterpret_cast<uint4*>(dyd + (((size_t)b * (H + 2) + 2 * y + 1) * (W + 2) + 2 * x + 1) * C + (size_t)c8 * 8) =
            *reinterpret_cast<const uint4*>(dy + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + (size_t)c8 * 8);
    }
}
// inverse of the ConvTranspose pixel-shuffle scatter (MAP_PIXSHUF): dL padded [B, gh*s+2, gw*s+2, C] -> dG rows
// [B*gh*gw, s*s*C], dG[(b,y,x), (i*s+j)*C + co] = dL[b, y*s+i+1, x*s+j+1, co]
__global__ void unpixshuf_kernel(const uint16_t* __restrict__ dl, uint16_t* __restrict__ dg, int B, int gh, int gw, int s, int C) {
    const int c8n = C / 8;
    const size_t total = (size_t)B * gh * gw * s * s * c8n;
    const int Wd = gw * s + 2, Hd = gh * s + 2;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % c8n);
        size_t p = idx / c8n;
        const int ij = (int)(p % (s * s)); p /= (s * s);
        const int x = (int)(p % gw); p /= gw;
        const int y = (int)(p % gh);
        const int b = (int)(p / gh);
        const int i = ij / s, j = ij - i * s;
        reinterpret_cast<uint4*>(dg)[idx] =
            *reinterpret_cast<const uint4*>(dl + (((size_t)b * Hd + y * s + i + 1) * Wd + x * s + j + 1) * C + (size_t)c8 * 8);
    }
}
// backward of the ProjectReadout concat (lseg_vit.py:87-88): d_cat [B*np, 2D] (16-bit) accumulated into the fp32 gradient of
// the hooked activation gx [B, ntok, D]:  gx[b, 1+p, :] += d_cat[b*np+p, :D] ;  gx[b, 0, :] += sum_p d_cat[b*np+p, D:]
__global__ void readout_cat_bwd_kernel(const uint16_t* __restrict__ dcat, float* __restrict__ gx, int B, int ntok, int D, int dtype) {
    const int np = ntok - 1, d8n = D >> 3;                // patch rows: 8 columns (16 bytes of d_cat, 32 bytes of gx) per lane
    const size_t total = (size_t)B * np * d8n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d0 = (int)(idx % d8n) * 8;
        const size_t bp = idx / d8n;
        const int p = (int)(bp % np), b = (int)(bp / np);
        const uint4 u = *reinterpret_cast<const uint4*>(dcat + bp * 2 * D + d0);
        const uint16_t* e = reinterpret_cast<const uint16_t*>(&u);
        float4* o = reinterpret_cast<float4*>(gx + ((size_t)b * ntok + 1 + p) * D + d0);
        float4 a0 = o[0], a1 = o[1];
        a0.x += load_as_f32(e, 0, dtype); a0.y += load_as_f32(e, 1, dtype); a0.z += load_as_f32(e, 2, dtype); a0.w += load_as_f32(e, 3, dtype);
        a1.x += load_as_f32(e, 4, dtype); a1.y += load_as_f32(e, 5, dtype); a1.z += load_as_f32(e, 6, dtype); a1.w += load_as_f32(e, 7, dtype);
        o[0] = a0; o[1] = a1;
    }
}
// the cls row: one block per (image, 256 columns) = 32 column groups x 32 row lanes, fixed summation order (no atomics)
__global__ __launch_bounds__(1024) void readout_cls_bwd_kernel(const uint16_t* __restrict__ dcat, float* __restrict__ gx, int ntok, int D, int dtype) {
    __shared__ float red[32][32][9];
    const int np = ntok - 1, b = blockIdx.y;
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + cg) * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < D) {
        for (int p = rl; p < np; p += 32) {
            const uint4 u = *reinterpret_cast<const uint4*>(dcat + ((size_t)b * np + p) * 2 * D + D + c0);
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&u);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += load_as_f32(e, k, dtype);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl][cg][k] = s[k];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int k = threadIdx.x & 7, g = threadIdx.x >> 3;
        const int c = (blockIdx.x * 32 + g) * 8 + k;
        if (c < D) {
            float t = 0.f;
            for (int j = 0; j < 32; ++j) t += red[j][g][k];
            gx[(size_t)b * ntok * D + c] += t;
        }
    }
}
// gradient of the embedding stage (lseg_vit.py:179-193) from gx = d x_0 [B, ntok, D] fp32:
//   dtok rows [B*np, D] 16-bit (the patch-embed GEMM's dY)  and  dpos_sum [ntok, D] fp32 = sum_b gx[b]  (row 0 = d cls_token)
__global__ void embed_bwd_kernel(const float* __restrict__ gx, uint16_t* __restrict__ dtok, float* __restrict__ dpos, int B, int ntok,
                                 int D, int dtype) {
    const size_t total = (size_t)ntok * D;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % D), t = (int)(idx / D);
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float v = gx[((size_t)b * ntok + t) * D + d];
            acc += v;
            if (t > 0) store_from_f32(dtok, ((size_t)b * (ntok - 1) + t - 1) * D + d, dtype, v);
        }
        dpos[idx] = acc;
    }
}
// transpose of pos_resize_kernel: dpos [1 + gh*gw, D] -> d pos_embed [1 + g_old^2, D] (+=; the caller zeroes unless accumulating), plus
// d cls_token = dpos[0] (cls_token is added to row 0 of every image).  GATHER form: a thread owns one (old cell, channel) and sums the
// resized positions whose bilinear footprint touches it in a fixed order (y ascending, x ascending) -- no atomics, bit-reproducible.
__global__ void pos_resize_bwd_kernel(const float* __restrict__ dpos, float* __restrict__ dposemb, float* __restrict__ dcls, int g_old,
                                      int gh, int gw, int D) {
    const size_t total = (size_t)(1 + g_old * g_old) * D;
    const float ry = (float)g_old / (float)gh, rx = (float)g_old / (float)gw;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % D);
        const int t = (int)(idx / D);
        if (t == 0) { const float g = dpos[d]; dposemb[d] += g; if (dcls) dcls[d] += g; continue; }
        const int oy = (t - 1) / g_old, ox = (t - 1) - oy * g_old;
        float acc = 0.f;
        for (int y = 0; y < gh; ++y) {
            const float sy = fmaxf(0.f, ((float)y + 0.5f) * ry - 0.5f);
            const int y0 = (int)sy, y1 = y0 + (y0 < g_old - 1);
            if (y0 != oy && y1 != oy) continue;
            const float ly = sy - (float)y0;
            const float wy = (y0 == oy ? 1.f - ly : 0.f) + (y1 == oy ? ly : 0.f);
            for (int x = 0; x < gw; ++x) {
                const float sx = fmaxf(0.f, ((float)x + 0.5f) * rx - 0.5f);
                const int x0 = (int)sx, x1 = x0 + (x0 < g_old - 1);
                if (x0 != ox && x1 != ox) continue;
                const float lx = sx - (float)x0;
                const float wx = (x0 == ox ? 1.f - lx : 0.f) + (x1 == ox ? lx : 0.f);
                acc += wy * wx * dpos[((size_t)1 + (size_t)y * gw + x) * D + d];
            }
        }
        dposemb[idx] += acc;
    }
}
// gradient re-layouts into the reference's parameter shapes (dst = or += src):
//   conv3x3  : dw tap-major [Co_p, 9, Ci_p] -> OIHW [Co, Ci, 3, 3]
//   convT k=s: dw [(i*s+j)*Cp + co, Cp(ci)] -> [Ci, Co, s, s]          (ConvTranspose2d weight layout, lseg_vit.py:457-466)
//   fold     : dst[c] (+)= sum_r src[r*ld + c], r < R                    (ConvTranspose bias: the s*s column groups share one bias)
__global__ void conv_wgrad_unpack_kernel(const float* __restrict__ dw, float* __restrict__ dst, int Co, int Ci, int Cip, int accumulate,
                                         int nsplit, size_t split_stride) {
    const size_t n = (size_t)Co * Ci * 9;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % 9), ci = (int)((i / 9) % Ci), co = (int)(i / ((size_t)9 * Ci));
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += dw[(size_t)s * split_stride + ((size_t)co * 9 + t) * Cip + ci];     // split-K partials
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
__global__ void convT_wgrad_unpack_kernel(const float* __restrict__ dw, float* __restrict__ dst, int C, int Cp, int s, int accumulate) {
    const size_t n = (size_t)C * C * s * s;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % s), i = (int)((idx / s) % s), co = (int)((idx / ((size_t)s * s)) % C), ci = (int)(idx / ((size_t)s * s * C));
        const float v = dw[((size_t)(i * s + j) * Cp + co) * Cp + ci];
        dst[idx] = accumulate ? dst[idx] + v : v;
    }
}
__global__ void fold_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C, int ld, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += src[(size_t)r * ld + c];
    dst[c] = accumulate ? dst[c] + s : s;
}
// running statistics of train-mode BatchNorm (nn.BatchNorm2d / SyncBatchNorm, momentum 0.1): updated in the caller's tensors
__global__ void bn_running_update_kernel(const float* __restrict__ stats, float* __restrict__ rmean, float* __restrict__ rvar, int C,
                                         float inv_n, float unbias, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c] * inv_n;
    const float var = fmaxf(stats[C + c] * inv_n - mean * mean, 0.f);
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * unbias;
}
// SGD with momentum + weight decay (torch.optim.SGD semantics, lsegmentation_module.py:165-171), fused with the refresh of the
// engine's packed 16-bit copy when the parameter is used as is (Linear weights):  g += wd*w ; m = mu*m + g ; w -= lr*m
__global__ void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, uint16_t* __restrict__ w16,
                           size_t n, float lr, float mu, float wd, int first, int dtype) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        const float gi = g[i] + wd * wi;
        const float mi = first ? gi : mu * m[i] + gi;       // torch: the momentum buffer starts as a copy of the first gradient
        m[i] = mi;
        const float wn = wi - lr * mi;
        w[i] = wn;
        if (w16) store_from_f32(w16, i, dtype, wn);
    }
}

// the whole optimizer step in ONE launch: block -> (parameter, 4096-element chunk) through a table sorted by first block.  Besides the
// master / momentum update it refreshes the engine's same-layout copies of the parameter (16-bit MFMA operand and / or fp32 copy).
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdSeg* __restrict__ segs, int nseg, float lr_pre, float lr_scr, float mu, float wd,
                                                        int first, int dtype) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SgdSeg sg = segs[lo];
    const float lr = sg.scratch ? lr_scr : lr_pre;
    const size_t base = (size_t)(blockIdx.x - sg.blk0) * 4096;
    if (sg.vec) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const size_t e = base + (size_t)it * 1024 + threadIdx.x * 4;
            if (e + 4 <= sg.n) {
                const float4 wi = *reinterpret_cast<const float4*>(sg.w + e), gi = *reinterpret_cast<const float4*>(sg.g + e);
                float4 mi = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!first) mi = *reinterpret_cast<const float4*>(sg.m + e);
                const float g0 = gi.x + wd * wi.x, g1 = gi.y + wd * wi.y, g2 = gi.z + wd * wi.z, g3 = gi.w + wd * wi.w;
                mi.x = first ? g0 : mu * mi.x + g0; mi.y = first ? g1 : mu * mi.y + g1;
                mi.z = first ? g2 : mu * mi.z + g2; mi.w = first ? g3 : mu * mi.w + g3;
                const float4 wn = make_float4(wi.x - lr * mi.x, wi.y - lr * mi.y, wi.z - lr * mi.z, wi.w - lr * mi.w);
                *reinterpret_cast<float4*>(sg.m + e) = mi;
                *reinterpret_cast<float4*>(sg.w + e) = wn;
                if (sg.w32) *reinterpret_cast<float4*>(sg.w32 + e) = wn;
                if (sg.w16) *reinterpret_cast<uint2*>(sg.w16 + e) = make_uint2(pack2_dt(wn.x, wn.y, dtype), pack2_dt(wn.z, wn.w, dtype));
            }
        }
        return;
    }
    for (int it = 0; it < 16; ++it) {
        const size_t e = base + (size_t)it * 256 + threadIdx.x;
        if (e < sg.n) {
            const float wi = sg.w[e];
            const float gi = sg.g[e] + wd * wi;
            const float mi = first ? gi : mu * sg.m[e] + gi;       // torch: the momentum buffer starts as a copy of the first gradient
            sg.m[e] = mi;
            const float wn = wi - lr * mi;
            sg.w[e] = wn;
            if (sg.w32) sg.w32[e] = wn;
            if (sg.w16) store_from_f32(sg.w16, e, dtype, wn);
        }
    }
}

// many small buffers zeroed in one launch (the atomically accumulated sums of a step: bias gradients, BatchNorm batch sums)
__global__ __launch_bounds__(256) void zero_multi_kernel(const ZeroJob* __restrict__ jobs, int njobs) {
    const ZeroJob j = jobs[blockIdx.x];
    for (unsigned i = threadIdx.x; i < j.n; i += 256) j.p[i] = 0.f;
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;      // cap + grid-stride (cdna guide G11)
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define CHECK_LAUNCH() LSEG_HIP_TRY(hipGetLastError())

int launch_layernorm(const void* in, int in_dtype, const float* gamma, const float* beta, void* out, int out_dtype,
                     int M, int D, float eps, hipStream_t st) {
    if (D % 4 != 0 || D > 64 * 4 * 8) return set_error(LSEG_ERR_UNSUPPORTED, "layernorm: D=%d", D);
    const int blocks = (M + 3) / 4;
    if (D <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, st, in, in_dtype, gamma, beta, out, out_dtype, M, D, eps);
    else if (D <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, st, in, in_dtype, gamma, beta, out, out_dtype, M, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, dim3(blocks), dim3(256), 0, st, in, in_dtype, gamma, beta, out, out_dtype, M, D, eps);
    else hipLaunchKernelGGL(layernorm_kernel<8>, dim3(blocks), dim3(256), 0, st, in, in_dtype, gamma, beta, out, out_dtype, M, D, eps);
    CHECK_LAUNCH();
    return 0;
}
int launch_conv_reduce_pad(const float* part, int ns, size_t stride, const float* bias, const void* res, const void* res2, void* out, void* out_relu,
                           int B, int Ho, int Wo, int C, int relu, int dtype, hipStream_t st) {
    if (C % 8) return set_error(LSEG_ERR_UNSUPPORTED, "conv_reduce_pad: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(conv_reduce_pad_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 8))), dim3(256), 0, st, part, ns, stride, bias,
                       (const uint16_t*)res, (const uint16_t*)res2, (uint16_t*)out, (uint16_t*)out_relu, B, Ho, Wo, C, relu, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_layernorm_reduce(float* x, const float* part, int nsplit, size_t stride, const float* bias, const float* gamma, const float* beta,
                            void* out, int out_dtype, int M, int D, float eps, hipStream_t st) {
    if (D % 4 != 0 || D > 64 * 4 * 8 || out_dtype == DT_F32) return set_error(LSEG_ERR_UNSUPPORTED, "layernorm_reduce: D=%d / 16-bit output only", D);
    const int blocks = (M + 3) / 4;
#define LNR(V) hipLaunchKernelGGL(layernorm_reduce_kernel<V>, dim3(blocks), dim3(256), 0, st, x, part, nsplit, stride, bias, gamma, beta, out, out_dtype, M, D, eps)
    if (D <= 256) LNR(1); else if (D <= 512) LNR(2); else if (D <= 1024) LNR(4); else LNR(8);
#undef LNR
    CHECK_LAUNCH();
    return 0;
}
int launch_im2col_patch(const float* x, void* A, int B, int H, int W, int P, int dtype, hipStream_t st) {
    const size_t total = (size_t)B * (H / P) * (W / P) * (3 * P * P / 8);
    hipLaunchKernelGGL(im2col_patch_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, (uint16_t*)A, B, H, W, P, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_pos_resize(const float* pos, float* out, int g_old, int gh, int gw, int D, hipStream_t st) {
    hipLaunchKernelGGL(pos_resize_kernel, dim3(grid_for((size_t)(1 + gh * gw) * D)), dim3(256), 0, st, pos, out, g_old, gh, gw, D);
    CHECK_LAUNCH();
    return 0;
}
int launch_cls_rows(const float* cls, const float* pos, float* x, int B, int ntok, int D, hipStream_t st) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, cls, pos, x, B, ntok, D);
    CHECK_LAUNCH();
    return 0;
}
int launch_readout_cat(const float* x, void* A, int B, int ntok, int D, int dtype, hipStream_t st) {
    const size_t total = (size_t)B * (ntok - 1) * (2 * D / 8);
    hipLaunchKernelGGL(readout_cat_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, (uint16_t*)A, B, ntok, D, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample2x_nhwc(const void* in, void* out, int B, int H, int W, int C, int dtype, hipStream_t st) {
    if (C % 8) return set_error(LSEG_ERR_UNSUPPORTED, "upsample2x_nhwc: C=%d", C);
    const size_t total = (size_t)B * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, W, C, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample2x_planes(const float* in, float* out, int P, int H, int W, hipStream_t st) {
    if (W % 4) return set_error(LSEG_ERR_UNSUPPORTED, "upsample2x_planes: W=%d must be a multiple of 4", W);
    const size_t blocks = (size_t)P * H;
    if (blocks > 0x7fffffffu) return set_error(LSEG_ERR_UNSUPPORTED, "upsample2x_planes: too many rows");
    hipLaunchKernelGGL(upsample2x_planes_kernel, dim3((unsigned)blocks), dim3(128), 2 * W * sizeof(float), st, in, out, P, H, W);
    CHECK_LAUNCH();
    return 0;
}
int launch_l2norm_scale_f16(const float* f, void* a, int M, int C, float scale, hipStream_t st) {
    if (C % 4 != 0 || C > 64 * 4 * 4) return set_error(LSEG_ERR_UNSUPPORTED, "l2norm: C=%d", C);
    const int blocks = (M + 3) / 4;
    if (C <= 256) hipLaunchKernelGGL(l2norm_scale_f16_kernel<1>, dim3(blocks), dim3(256), 0, st, f, (uint16_t*)a, M, C, scale);
    else if (C <= 512) hipLaunchKernelGGL(l2norm_scale_f16_kernel<2>, dim3(blocks), dim3(256), 0, st, f, (uint16_t*)a, M, C, scale);
    else hipLaunchKernelGGL(l2norm_scale_f16_kernel<4>, dim3(blocks), dim3(256), 0, st, f, (uint16_t*)a, M, C, scale);
    CHECK_LAUNCH();
    return 0;
}
int launch_combine_1x1(const float* wh, const float* bh, const float* wo, const float* bo, float* wc, float* bc, int Co, int Cm, int Ci,
                       hipStream_t st) {
    const int n = Co * (Ci + 1);
    hipLaunchKernelGGL(combine_1x1_kernel, dim3((n + 255) / 256), dim3(256), 0, st, wh, bh, wo, bo, wc, bc, Co, Cm, Ci);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample_norm_f16(const float* g, void* a, int B, int H, int W, int C, float scale, hipStream_t st) {
    if (C % 8 != 0 || C > 1024) return set_error(LSEG_ERR_UNSUPPORTED, "upsample_norm: C=%d", C);
    const size_t nblk = (size_t)B * H * W;               // 2x2 output blocks, one per wave
    const int blocks = (int)std::min<size_t>((nblk + 3) / 4, 256 * 32);
    if (C <= 512) hipLaunchKernelGGL(upsample_norm_f16_kernel<1>, dim3(blocks), dim3(256), 0, st, g, (uint16_t*)a, B, H, W, C, scale);
    else hipLaunchKernelGGL(upsample_norm_f16_kernel<2>, dim3(blocks), dim3(256), 0, st, g, (uint16_t*)a, B, H, W, C, scale);
    CHECK_LAUNCH();
    return 0;
}
int launch_text_embed(const int64_t* tok, const float* emb, const float* pos, void* x, int rows, int L, int ctx, int W, hipStream_t st) {
    hipLaunchKernelGGL(text_embed_kernel, dim3(grid_for((size_t)rows * W)), dim3(256), 0, st, tok, emb, pos, (uint16_t*)x, rows, L, ctx, W);
    CHECK_LAUNCH();
    return 0;
}
int launch_text_pool(const void* x, const int* eot, void* pooled, int K, int L, int W, hipStream_t st) {
    hipLaunchKernelGGL(text_pool_kernel, dim3((K * W + 255) / 256), dim3(256), 0, st, (const uint16_t*)x, eot, (uint16_t*)pooled, K, L, W);
    CHECK_LAUNCH();
    return 0;
}
int launch_text_l2norm(const void* t, void* out, int K, int C, hipStream_t st) {
    hipLaunchKernelGGL(text_l2norm_kernel, dim3((K + 3) / 4), dim3(256), 0, st, (const uint16_t*)t, (uint16_t*)out, K, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_range16(const void* p, size_t n, int dtype, unsigned long long* out4, hipStream_t st) {
    if (!n) return 0;
    const size_t work = (n >> 3) + (n & 7);
    unsigned blocks = (unsigned)std::min<size_t>((work + 255) / 256, 4096);
    hipLaunchKernelGGL(range16_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)p, n, dtype, out4);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_convert(const void* in, int in_dtype, void* out, int out_dtype, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(convert_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, in_dtype, out, out_dtype, n);
    CHECK_LAUNCH();
    return 0;
}
int launch_convert2d(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, int ld, hipStream_t st) {
    hipLaunchKernelGGL(convert2d_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, st, in, in_dtype, out, out_dtype, R, C, ld);
    CHECK_LAUNCH();
    return 0;
}
int launch_transpose_convert(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, hipStream_t st) {
    hipLaunchKernelGGL(transpose_convert_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, st, in, in_dtype, out, out_dtype, R, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_pack_conv3x3(const float* w, const float* bn_w, const float* bn_b, const float* bn_m, const float* bn_v,
                        float bn_eps, const float* conv_bias, void* wp, float* bias_out, int Co, int Ci, int Cip, int dtype,
                        hipStream_t st) {
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for((size_t)Co * 9 * Ci)), dim3(256), 0, st, w, bn_w, bn_b, bn_m, bn_v,
                       bn_eps, conv_bias, wp, bias_out, Co, Ci, Cip, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_pack_convT(const float* w, void* wp, int Ci, int Co, int Cp, int s, int dtype, hipStream_t st) {
    hipLaunchKernelGGL(pack_convT_kernel, dim3(grid_for((size_t)s * s * Co * Ci)), dim3(256), 0, st, w, wp, Ci, Co, Cp, s, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_nhwc_to_nchw_f32(const void* in, float* out, int B, int H, int W, int C, int Cs, int pad, int dtype, hipStream_t st, size_t lo_plane) {
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, st, (const uint16_t*)in, out, B, H, W, C, Cs, pad, dtype, lo_plane);
    CHECK_LAUNCH();
    return 0;
}
int launch_rows_to_nchw_f32(const float* in, float* out, int B, int HW, int C, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_nchw_f32_kernel, dim3(grid_for((size_t)B * C * HW)), dim3(256), 0, st, in, out, B, HW, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_head_block(const float* in, float* out, const float* w9, const float* bias, int B, int K, int H, int W,
                      int bottleneck, int act, int apply_act, hipStream_t st) {
    hipLaunchKernelGGL(head_block_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, in, out, w9, bias, B, K, H, W, bottleneck, act, apply_act);
    CHECK_LAUNCH();
    return 0;
}
int launch_argmax_planes(const float* in, uint8_t* out, int B, int K, int HW, hipStream_t st) {
    hipLaunchKernelGGL(argmax_planes_kernel, dim3(grid_for((size_t)B * HW)), dim3(256), 0, st, in, out, B, K, HW);
    CHECK_LAUNCH();
    return 0;
}

int launch_layernorm_backward(const void* dy, int dy_dtype, const float* x, const float* gamma, float* dx, float* dgamma,
                              float* dbeta, int M, int D, float eps, int accumulate, hipStream_t st, int accumulate_params,
                              float* partial_ws, void* dx16) {
    if (D % 4 != 0 || D > 64 * 4 * 4) return set_error(LSEG_ERR_UNSUPPORTED, "layernorm backward: D=%d", D);
    if (dx16 && dy_dtype == DT_F32) return set_error(LSEG_ERR_INVALID, "layernorm backward: the 16-bit copy takes dy's 16-bit type");
    int blocks = (M + 3) / 4;
    if (partial_ws) {
        if (blocks > LN_BWD_PARTIAL_BLOCKS) blocks = LN_BWD_PARTIAL_BLOCKS;
    } else {
        if (blocks > 1024) blocks = 1024;
        if (!accumulate_params) {
            LSEG_HIP_TRY(hipMemsetAsync(dgamma, 0, (size_t)D * sizeof(float), st));
            LSEG_HIP_TRY(hipMemsetAsync(dbeta, 0, (size_t)D * sizeof(float), st));
        }
    }
#define LN_BWD(V) hipLaunchKernelGGL(layernorm_bwd_kernel<V>, dim3(blocks), dim3(256), 0, st, dy, dy_dtype, x, gamma, dx, dgamma, dbeta, M, D, eps, accumulate, partial_ws, (uint16_t*)dx16)
    if (D <= 256) LN_BWD(1); else if (D <= 512) LN_BWD(2); else LN_BWD(4);
#undef LN_BWD
    CHECK_LAUNCH();
    if (partial_ws) {
        if ((((uintptr_t)dgamma | (uintptr_t)dbeta | (uintptr_t)partial_ws) & 15) == 0)
            hipLaunchKernelGGL(colreduce_kernel<16>, dim3((2 * D + 15) / 16), dim3(1024), 0, st, partial_ws, dgamma, dbeta, blocks, D, D, 2 * D, accumulate_params);
        else
            hipLaunchKernelGGL(colreduce_scalar_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, st, partial_ws, dgamma, dbeta, blocks, D, D, 2 * D, accumulate_params);
        CHECK_LAUNCH();
    }
    return 0;
}

int launch_transpose16(const void* in, void* out, int R, int C, int ldi, int ldo, hipStream_t st, int shift, int relu) {
    dim3 grid((ldo + 63) / 64, (C + 63) / 64);
    if (!(C & 7) && !(ldi & 7) && !(ldo & 7) && !((uintptr_t)in & 15) && !((uintptr_t)out & 15))
        hipLaunchKernelGGL(transpose16_vec_kernel, grid, dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, R, C, ldi, ldo, shift, relu);
    else
        hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, R, C, ldi, ldo, shift, relu);
    CHECK_LAUNCH();
    return 0;
}
int launch_conv_dgrad_pack(const void* wp, void* wd, int Co, int Ci, hipStream_t st) {
    hipLaunchKernelGGL(conv_dgrad_pack_kernel, dim3(grid_for((size_t)Co * 9 * Ci)), dim3(256), 0, st, (const uint16_t*)wp, (uint16_t*)wd, Co, Ci);
    CHECK_LAUNCH();
    return 0;
}
int launch_gelu_backward(const void* dy, const void* pre, void* dx, size_t n, int dtype, hipStream_t st, int quick) {
    if (!(n & 7) && !(((uintptr_t)dy | (uintptr_t)pre | (uintptr_t)dx) & 15))
        hipLaunchKernelGGL(gelu_backward_vec_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)pre, (uint16_t*)dx, n / 8, dtype, quick);
    else
        hipLaunchKernelGGL(gelu_backward_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)pre, (uint16_t*)dx, n, dtype, quick);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample2x_nhwc_backward(const void* dout, void* din, int B, int H, int W, int C, int dtype, hipStream_t st) {
    if (C % 8) return set_error(LSEG_ERR_UNSUPPORTED, "upsample2x_nhwc backward: C=%d", C);
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_nhwc_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const uint16_t*)dout, (uint16_t*)din, B, H, W, C, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_softmax_ce_backward(const float* scores, const int64_t* target, float* dz, int B, int K, int HW, int ignore_index,
                               const double* nll, hipStream_t st) {
    const size_t npix = (size_t)B * HW;
    hipLaunchKernelGGL(softmax_ce_backward_kernel, dim3(grid_for(npix)), dim3(256), 0, st, scores,
                       reinterpret_cast<const long long*>(target), dz, K, HW, npix, ignore_index, nll);
    CHECK_LAUNCH();
    return 0;
}
// BatchNorm in train mode, split so that a SyncBatchNorm exchange (all-reduce of the [2C] sums) can sit between the two halves.
// `count` = pixels behind the sums (B*H*W, times the world size after the exchange).
// rows per block of the column-statistics kernels: ~1500 blocks over the chip, at least 32 rows (4 per row lane) per block
static int colstats_rows(int R, int C) {
    const int cb = (C + 255) / 256;
    int rpb = (int)(((long)R * cb + 1499) / 1500);
    rpb = (rpb + 7) / 8 * 8;
    return rpb < 32 ? 32 : rpb;
}
// deterministic finish of a column-statistics launch: out[0..ncol) (+)= the nb partial rows, fixed order
static int colstats_reduce(const float* partial, float* out, int nb, int ncol, int accumulate, hipStream_t st) {
    if (!(ncol & 3) && !(((uintptr_t)out | (uintptr_t)partial) & 15))
        hipLaunchKernelGGL(colreduce_kernel<16>, dim3((ncol + 15) / 16), dim3(1024), 0, st, partial, out, (float*)nullptr, nb, ncol, 0, ncol, accumulate);
    else
        hipLaunchKernelGGL(colreduce_scalar_kernel, dim3((ncol + 63) / 64), dim3(1024), 0, st, partial, out, (float*)nullptr, nb, ncol, 0, ncol, accumulate);
    CHECK_LAUNCH();
    return 0;
}
// rows per block such that the partial rows of a deterministic launch fit `cap` floats
static int colstats_rows_det(int R, int C, int ncol, size_t cap) {
    int rpb = colstats_rows(R, C);
    const size_t max_rows = cap / (size_t)ncol;
    if (max_rows == 0) return -1;
    if ((size_t)((R + rpb - 1) / rpb) > max_rows) rpb = (int)(((size_t)R + max_rows - 1) / max_rows + 7) / 8 * 8;
    return rpb;
}
int launch_bn_stats(const void* x, float* stats, int B, int H, int W, int C, int dtype, hipStream_t st, int pre_zeroed, float* det_ws, size_t det_cap) {
    const int Mp = B * (H + 2) * (W + 2);
    const int rpb = det_ws ? colstats_rows_det(Mp, C, 2 * C, det_cap) : colstats_rows(Mp, C);
    if (rpb < 0) return set_error(LSEG_ERR_INVALID, "bn_stats: deterministic workspace too small for C=%d", C);
    if (!pre_zeroed && !det_ws) LSEG_HIP_TRY(hipMemsetAsync(stats, 0, (size_t)2 * C * sizeof(float), st));
    const int nb = (Mp + rpb - 1) / rpb;
    hipLaunchKernelGGL(colstats16_kernel<1>, dim3((C + 255) / 256, nb), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)nullptr,
                       (const float*)nullptr, 0.f, 0.f, dtype, stats, Mp, C, C, rpb, det_ws);
    CHECK_LAUNCH();
    return det_ws ? colstats_reduce(det_ws, stats, nb, 2 * C, 0, st) : 0;      // (a pre-zeroed buffer is simply overwritten)
}
int launch_bn_apply(const void* x, void* y, const float* stats, const float* gamma, const float* beta, const void* res1, const void* res2,
                    int B, int H, int W, int C, float eps, double count, int dtype, hipStream_t st, void* y_relu) {
    if (C % 8) return set_error(LSEG_ERR_UNSUPPORTED, "batch norm: C=%d must be a multiple of 8", C);
    if (!y && !y_relu) return set_error(LSEG_ERR_INVALID, "batch norm: no output");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for((size_t)B * H * W * (C / 8))), dim3(256), 0, st, (const uint16_t*)x, (uint16_t*)y,
                       stats, gamma, beta, B, H, W, C, eps, dtype, (float)(1.0 / count), (const uint16_t*)res1, (const uint16_t*)res2,
                       (uint16_t*)y_relu);
    CHECK_LAUNCH();
    return 0;
}
int launch_bn_bwd_stats(const void* dy, const void* x, const float* stats, float* bstats, int B, int H, int W, int C, float eps,
                        double count, int dtype, hipStream_t st, float* det_ws, size_t det_cap) {
    const int Mp = B * (H + 2) * (W + 2);
    const int rpb = det_ws ? colstats_rows_det(Mp, C, 2 * C, det_cap) : colstats_rows(Mp, C);
    if (rpb < 0) return set_error(LSEG_ERR_INVALID, "bn_bwd_stats: deterministic workspace too small for C=%d", C);
    if (!det_ws) LSEG_HIP_TRY(hipMemsetAsync(bstats, 0, (size_t)2 * C * sizeof(float), st));
    const int nb = (Mp + rpb - 1) / rpb;
    hipLaunchKernelGGL(colstats16_kernel<2>, dim3((C + 255) / 256, nb), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x,
                       stats, (float)(1.0 / count), eps, dtype, bstats, Mp, C, C, rpb, det_ws);
    CHECK_LAUNCH();
    return det_ws ? colstats_reduce(det_ws, bstats, nb, 2 * C, 0, st) : 0;
}
int launch_bn_bwd_apply(const void* dy, const void* x, const float* stats, const float* bstats, const float* gamma, void* dx,
                        int B, int H, int W, int C, float eps, double count, int dtype, hipStream_t st) {
    if (C % 8) return set_error(LSEG_ERR_UNSUPPORTED, "batch norm: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)B * H * W * (C / 8))), dim3(256), 0, st, (const uint16_t*)dy,
                       (const uint16_t*)x, (uint16_t*)dx, stats, bstats, gamma, B, H, W, C, eps, dtype, (float)(1.0 / count));
    CHECK_LAUNCH();
    return 0;
}
int launch_bn_train_forward(const void* x, void* y, float* stats, const float* gamma, const float* beta, int B, int H, int W, int C,
                            float eps, int dtype, hipStream_t st) {
    int r = launch_bn_stats(x, stats, B, H, W, C, dtype, st);
    if (r || !y) return r;
    return launch_bn_apply(x, y, stats, gamma, beta, nullptr, nullptr, B, H, W, C, eps, (double)B * H * W, dtype, st);
}
int launch_bn_train_backward(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, float* bstats,
                             int B, int H, int W, int C, float eps, int dtype, hipStream_t st) {
    const double count = (double)B * H * W;
    int r = launch_bn_bwd_stats(dy, x, stats, bstats, B, H, W, C, eps, count, dtype, st);
    if (r) return r;
    return launch_bn_bwd_apply(dy, x, stats, bstats, gamma, dx, B, H, W, C, eps, count, dtype, st);
}
int launch_relu_backward(const void* dy, const void* x, void* dx, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(relu_backward_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x, (uint16_t*)dx, n);
    CHECK_LAUNCH();
    return 0;
}
int launch_relu_backward_add(const void* dy, const void* x, const void* add, void* dx, size_t n, int dtype, hipStream_t st) {
    hipLaunchKernelGGL(relu_backward_add_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x,
                       (const uint16_t*)add, (uint16_t*)dx, n, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample2x_planes_backward_rows(const float* dout, void* rows, int B, int K, int H, int W, int ldk, int dtype, hipStream_t st) {
    if (ldk < K) return set_error(LSEG_ERR_INVALID, "upsample2x_planes backward: ldk=%d < K=%d", ldk, K);
    hipLaunchKernelGGL(upsample2x_planes_bwd_rows_kernel, dim3(grid_for((size_t)B * K * H * W)), dim3(256), 0, st, dout, (uint16_t*)rows,
                       B, K, H, W, ldk, dtype);
    CHECK_LAUNCH();
    return 0;
}
// every output row Y whose bilinear footprint touches low row y lies in [2y-1, 2y+2] (same fp32 arithmetic as the kernels)
static bool x2_footprint_is_4(int H) {
    const int Ho = 2 * H;
    const float ry = (float)(H - 1) / (float)(Ho - 1);
    for (int yo = 0; yo < Ho; ++yo) {
        const float sy = ry * (float)yo;
        const int y0 = (int)sy, y1 = y0 + (y0 < H - 1);
        const float l = sy - (float)y0;
        if (1.f - l != 0.f && (yo < 2 * y0 - 1 || yo > 2 * y0 + 2)) return false;
        if (l != 0.f && y1 != y0 && (yo < 2 * y1 - 1 || yo > 2 * y1 + 2)) return false;
    }
    return true;
}
int launch_upsample_ce_backward_rows(const float* low, const int64_t* target, const float* lse, const double* nll, void* rows, int B, int K,
                                     int H, int W, int ldk, int ignore_index, int dtype, hipStream_t st, const float* gscale) {
    if (ldk < K || (ldk & 7)) return set_error(LSEG_ERR_INVALID, "fused CE backward: ldk=%d must be a multiple of 8 and >= K=%d", ldk, K);
    if (!x2_footprint_is_4(H) || !x2_footprint_is_4(W)) return set_error(LSEG_ERR_UNSUPPORTED, "fused CE backward: %dx%d map outside the 4-tap footprint", H, W);
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(upsample_ce_bwd_rows_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, low,
                       reinterpret_cast<const long long*>(target), lse, nll, (uint16_t*)rows, B, K, H, W, ldk, ignore_index, dtype, gscale);
    CHECK_LAUNCH();
    return 0;
}
int launch_l2norm_scale_backward(const void* da, int da_dtype, const float* x, void* dx, int dx_dtype, int M, int C, float scale, hipStream_t st) {
    if (C % 4 != 0 || C > 64 * 4 * 4) return set_error(LSEG_ERR_UNSUPPORTED, "l2norm backward: C=%d", C);
    const int blocks = (M + 3) / 4;
#define L2B(V) hipLaunchKernelGGL(l2norm_scale_bwd_kernel<V>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)da, da_dtype, x, (uint16_t*)dx, dx_dtype, M, C, scale)
    if (C <= 256) L2B(1); else if (C <= 512) L2B(2); else L2B(4);
#undef L2B
    CHECK_LAUNCH();
    return 0;
}
int launch_colsum16(const void* in, int dtype, float* out, int R, int C, int ld, hipStream_t st, int accumulate, float* det_ws, size_t det_cap) {
    if (!accumulate && !det_ws) LSEG_HIP_TRY(hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st));
    const bool vec = !(ld & 7) && !((uintptr_t)in & 15);
    const int rpb = det_ws ? colstats_rows_det(R, C, C, det_cap) : colstats_rows(R, C);
    if (rpb < 0) return set_error(LSEG_ERR_INVALID, "colsum: deterministic workspace too small for C=%d", C);
    dim3 grid((C + 255) / 256, (R + rpb - 1) / rpb);
    if (!vec) {          // odd leading dimensions (op-level tests): one thread per column and row chunk
        hipLaunchKernelGGL(colsum16_scalar_kernel, grid, dim3(256), 0, st, (const uint16_t*)in, dtype, out, R, C, ld, rpb, det_ws);
    } else {
        hipLaunchKernelGGL(colstats16_kernel<0>, grid, dim3(256), 0, st, (const uint16_t*)in, (const uint16_t*)nullptr, (const float*)nullptr, 0.f, 0.f,
                           dtype, out, R, C, ld, rpb, det_ws);
    }
    CHECK_LAUNCH();
    return det_ws ? colstats_reduce(det_ws, out, (int)grid.y, C, accumulate, st) : 0;
}

int launch_seg_stats(const float* scores, const int64_t* target, int B, int K, int HW, int ignore_index,
                     unsigned long long* counts, double* nll, hipStream_t st) {
    return launch_seg_stats_ex(scores, target, B, K, HW, ignore_index, counts, nll, nullptr, 0, 0, 0, st);
}
// up != 0: scores = low-resolution logits [B,K,h,w], statistics / masks on the x2-upsampled grid (HW = 4*h*w)
int launch_seg_stats_ex(const float* scores, const int64_t* target, int B, int K, int HW, int ignore_index, unsigned long long* counts,
                        double* nll, uint8_t* argmax_out, int up, int h, int w, hipStream_t st, float* lse_out) {
    if (K < 1 || K > 4096) return set_error(LSEG_ERR_INVALID, "seg_stats: K=%d", K);
    if (argmax_out && K > 256) return set_error(LSEG_ERR_UNSUPPORTED, "uint8 masks need K <= 256 (K=%d)", K);
    if (target) {
        LSEG_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)(2 + 3 * K) * sizeof(unsigned long long), st));
        LSEG_HIP_TRY(hipMemsetAsync(nll, 0, 2 * sizeof(double), st));
    }
    const size_t npix = (size_t)B * HW;
    int grid = (int)std::min<size_t>((npix + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(seg_stats_kernel, dim3(grid), dim3(256), (size_t)(3 * K + 2) * sizeof(unsigned int), st,
                       scores, reinterpret_cast<const long long*>(target), K, HW, npix, ignore_index, counts, nll, argmax_out, up, h, w, lse_out);
    CHECK_LAUNCH();
    return 0;
}


int launch_gelu_forward(const void* pre, void* out, size_t n, int dtype, hipStream_t st) {
    if (n % 8) return set_error(LSEG_ERR_UNSUPPORTED, "gelu forward: n=%zu must be a multiple of 8", n);
    hipLaunchKernelGGL(gelu_forward_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const uint16_t*)pre, (uint16_t*)out, n / 8, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_unpad_rows(const void* in, void* out, int B, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(unpad_rows_kernel, dim3(grid_for((size_t)B * H * W * (C / 8))), dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, W, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_dilate2(const void* dy, void* dyd, int B, int Ho, int Wo, int H, int W, int C, hipStream_t st) {
    LSEG_HIP_TRY(hipMemsetAsync(dyd, 0, (size_t)B * (H + 2) * (W + 2) * C * 2, st));
    hipLaunchKernelGGL(dilate2_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 8))), dim3(256), 0, st, (const uint16_t*)dy, (uint16_t*)dyd, B, Ho, Wo, H, W, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_unpixshuf(const void* dl, void* dg, int B, int gh, int gw, int s, int C, hipStream_t st) {
    hipLaunchKernelGGL(unpixshuf_kernel, dim3(grid_for((size_t)B * gh * gw * s * s * (C / 8))), dim3(256), 0, st, (const uint16_t*)dl, (uint16_t*)dg, B, gh, gw, s, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_readout_cat_bwd(const void* dcat, float* gx, int B, int ntok, int D, int dtype, hipStream_t st) {
    if (D % 8) return set_error(LSEG_ERR_UNSUPPORTED, "readout backward: D=%d must be a multiple of 8", D);
    hipLaunchKernelGGL(readout_cat_bwd_kernel, dim3(grid_for((size_t)B * (ntok - 1) * (D / 8))), dim3(256), 0, st, (const uint16_t*)dcat, gx, B, ntok, D, dtype);
    hipLaunchKernelGGL(readout_cls_bwd_kernel, dim3((D + 255) / 256, B), dim3(1024), 0, st, (const uint16_t*)dcat, gx, ntok, D, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_embed_bwd(const float* gx, void* dtok, float* dpos, int B, int ntok, int D, int dtype, hipStream_t st) {
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((size_t)ntok * D)), dim3(256), 0, st, gx, (uint16_t*)dtok, dpos, B, ntok, D, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_pos_resize_bwd(const float* dpos, float* dposemb, float* dcls, int g_old, int gh, int gw, int D, hipStream_t st) {
    hipLaunchKernelGGL(pos_resize_bwd_kernel, dim3(grid_for((size_t)(1 + g_old * g_old) * D)), dim3(256), 0, st, dpos, dposemb, dcls, g_old, gh, gw, D);
    CHECK_LAUNCH();
    return 0;
}
int launch_sum_partials(const float* part, float* dst, int nsplit, size_t n, size_t stride, int accumulate, hipStream_t st) {
    if ((n & 3) || (stride & 3)) return set_error(LSEG_ERR_INVALID, "sum_partials: n and stride must be multiples of 4");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, part, dst, nsplit, n / 4, stride / 4, accumulate);
    CHECK_LAUNCH();
    return 0;
}
int launch_conv_wgrad_unpack(const float* dw, float* dst, int Co, int Ci, int Cip, int accumulate, hipStream_t st, int nsplit, size_t split_stride) {
    hipLaunchKernelGGL(conv_wgrad_unpack_kernel, dim3(grid_for((size_t)Co * Ci * 9)), dim3(256), 0, st, dw, dst, Co, Ci, Cip, accumulate,
                       nsplit, split_stride);
    CHECK_LAUNCH();
    return 0;
}
int launch_convT_wgrad_unpack(const float* dw, float* dst, int C, int Cp, int s, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(convT_wgrad_unpack_kernel, dim3(grid_for((size_t)C * C * s * s)), dim3(256), 0, st, dw, dst, C, Cp, s, accumulate);
    CHECK_LAUNCH();
    return 0;
}
int launch_fold_rows(const float* src, float* dst, int R, int C, int ld, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(fold_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, st, src, dst, R, C, ld, accumulate);
    CHECK_LAUNCH();
    return 0;
}
int launch_bn_running_update(const float* stats, float* rmean, float* rvar, int C, double count, float momentum, hipStream_t st) {
    const float unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.0f;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, rmean, rvar, C, (float)(1.0 / count), unbias, momentum);
    CHECK_LAUNCH();
    return 0;
}
int launch_sgd(float* w, const float* g, float* m, void* w16, size_t n, float lr, float mu, float wd, int first, int dtype, hipStream_t st) {
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, st, w, g, m, (uint16_t*)w16, n, lr, mu, wd, first, dtype);
    CHECK_LAUNCH();
    return 0;
}

int launch_sgd_multi(const SgdSeg* dev_segs, int nseg, unsigned blocks, float lr_pre, float lr_scr, float mu, float wd, int first, int dtype,
                     hipStream_t st) {
    if (nseg < 1 || blocks < 1) return 0;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(blocks), dim3(256), 0, st, dev_segs, nseg, lr_pre, lr_scr, mu, wd, first, dtype);
    CHECK_LAUNCH();
    return 0;
}
int launch_transpose16_multi(const TransposeJob* dev_jobs, int njobs, unsigned blocks, hipStream_t st) {
    if (njobs < 1 || blocks < 1) return 0;
    hipLaunchKernelGGL(transpose16_multi_kernel, dim3(blocks), dim3(256), 0, st, dev_jobs, njobs);
    CHECK_LAUNCH();
    return 0;
}

// commuted correlation: cell dot products of g, the per-pixel scale plane, the scaled x2 upsample of the label planes
int launch_pixel_gram(const void* g16, float* gram, int B, int H, int W, int C, hipStream_t st) {
    if (C % 8 || C > 1024) return set_error(LSEG_ERR_UNSUPPORTED, "pixel gram: C=%d", C);
    const size_t nrun = (size_t)B * H * ((W + 3) / 4);
    const int blocks = (int)std::min<size_t>((nrun + 3) / 4, 256 * 32);
    if (C <= 512) hipLaunchKernelGGL(pixel_gram_kernel<1>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)g16, gram, B, H, W, C);
    else hipLaunchKernelGGL(pixel_gram_kernel<2>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)g16, gram, B, H, W, C);
    CHECK_LAUNCH();
    return 0;
}
int launch_norm_scale_plane(const float* gram, float* scale, int B, int H, int W, float s, hipStream_t st, unsigned* flag) {
    hipLaunchKernelGGL(norm_scale_plane_kernel, dim3(grid_for((size_t)B * 4 * H * W)), dim3(256), 0, st, gram, scale, B, H, W, s, flag);
    CHECK_LAUNCH();
    return 0;
}
int launch_upsample2x_planes_scaled(const float* in_padded, const float* scale, float* out, int P, int K, int H, int W, hipStream_t st) {
    if (W % 2 != 0) return set_error(LSEG_ERR_UNSUPPORTED, "scaled upsample: W=%d must be even", W);
    const int bands = (2 * H + UPS_LB - 1) / UPS_LB;
    hipLaunchKernelGGL(upsample2x_planes_scaled_kernel, dim3((unsigned)P * bands), dim3(256), (size_t)(UPS_LB / 2 + 4) * W * sizeof(float), st,
                       in_padded, scale, out, P, K, H, W);
    CHECK_LAUNCH();
    return 0;
}
// R planes -> the full-resolution logits in one pass (x2 with the per-pixel scale and fp16 rounding, then output_conv's x2)
int launch_upsample4x_planes_scaled(const float* in_padded, const float* scale, float* out, int P, int K, int H, int W, hipStream_t st) {
    if (W % 2 != 0) return set_error(LSEG_ERR_UNSUPPORTED, "scaled upsample: W=%d must be even", W);
    const int bands = (2 * H + UPS4_LB - 1) / UPS4_LB;
    const size_t lds = ((size_t)(UPS4_LB / 2 + 4) * W + (size_t)(UPS4_LB + 1) * 2 * W) * sizeof(float);
    hipLaunchKernelGGL((upsample4x_planes_scaled_kernel<UPS4_LB>), dim3((unsigned)P * bands), dim3(256), lds, st, in_padded, scale, out, P, K, H, W);
    CHECK_LAUNCH();
    return 0;
}

int launch_zero_multi(const ZeroJob* dev_jobs, int njobs, hipStream_t st) {
    if (njobs < 1) return 0;
    hipLaunchKernelGGL(zero_multi_kernel, dim3(njobs), dim3(256), 0, st, dev_jobs, njobs);
    CHECK_LAUNCH();
    return 0;
}

}  // namespace lseg

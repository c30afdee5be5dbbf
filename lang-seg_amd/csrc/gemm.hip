// gemm.hip -- bf16/fp16 MFMA GEMM + implicit-GEMM 3x3 conv for gfx950 (MI355X).
//
// One kernel template serves every dense contraction of the LSeg forward:
//   timm Linear layers  (qkv / proj / fc1 / fc2, [3P] timm vision_transformer.py Block)
//   patch embedding     (lseg_vit.py:179, after an im2col gather)
//   ProjectReadout      (lseg_vit.py:86-90), reassemble 1x1 convs + ConvTranspose (:446-523)
//   scratch 3x3 convs   (lseg_blocks.py:73-108, 237-255), out_conv / head1 1x1 convs
//   CLIP text Linear layers ([3P] clip/model.py ResidualAttentionBlock) in fp16
//   the pixel x text correlation (lseg_net.py:194) in fp16
//
// Structure: BMxBNx64 tiles (64x64, 128x128 with 4 waves; 256x256 with 8 waves; 64x512 for the fused head),
// each wave a 32x32 .. 64x128 sub-tile of v_mfma_f32_16x16x32 tiles; operands stream HBM -> LDS with direct-to-LDS
// loads (global_load_lds_dwordx4, 16 B/lane) into SEPARATE rings for A (3 K-steps) and W (2), counted vmcnt;
// software-pipelined K-loop with one barrier per K-step; XOR swizzle applied on the source address (LDS image
// stays lane-linear); persistent XCD-aware grouped tile rasterisation; compile-time epilogues for the hot shapes.
// The MFMA is issued "swapped" (weights as the row operand) so each lane ends up with 4 consecutive output
// channels of one row -> 8/16-byte epilogue stores and float4 bias/residual loads.  DESIGN.md §3.1 has the
// measurements behind each of these choices.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "gemm.h"
#include "../../include/lseg_hip.h"

namespace lseg {

// compute units of a device (cached per device id; 256 on MI355X): sizes the persistent grids
int device_cu_count(int dev) {
    static std::atomic<int> cache[64];
    const int slot = dev & 63;
    int v = cache[slot].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[slot].store(v, std::memory_order_relaxed);
    return v;
}

void gemm_args_init(GemmArgs& g) {
    g = GemmArgs{};
    g.stride = 1;
}

namespace {

// compile-time loop: keeps accumulator indices static (runtime-indexed register arrays go to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-roundoff class) -- ~12 VALU
// instead of the ~30 of ocml erff; GELU(erf) as in timm's nn.GELU.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-ax * ax);
    return x < 0.f ? -e : e;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// d/dz of the erf GELU: Phi(z) + z * phi(z); the exponential of the erf approximation IS exp(-z^2 / 2)
__device__ __forceinline__ float gelu_grad_as(float z) {
    const float ax = fabsf(z) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float ex = __expf(-ax * ax);
    const float e = 1.0f - poly * ex;
    return 0.5f * (1.0f + (z < 0.f ? -e : e)) + z * 0.3989422804014327f * ex;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// GELU(erf) through erfc(|z|) ~= exp2(|z| R(|z|)), z = x / sqrt 2: R = degree-5 fit of log2(erfc(z)) / z on [0, 4.4] (erfc(0) = 1 exactly; max
// |erfc error| 4.7e-7, GELU error <= 4.6e-7 absolute in fp32 evaluation -- tools/fit_gelu.py).  With h = erfc(|z|) / 2 = exp2(|z| R - 1):
//   y = x Phi(x) = max(x, 0) - |x| h        (x >= 0: x (1 - h);  x < 0: x h -- no cancellation in the negative tail)
// The 1 / sqrt 2 is folded into the coefficients (the polynomial runs in u = |x| directly: d_k = c_k 2^(-(k+1)/2)), |x| is a source
// modifier (the last fma is paired into a v_pk_fma_f32 by the SLP vectoriser, which costs a v_and per element): 11 VALU instructions, ONE
// transcendental -- against 13 + rcp + exp2 + two v_and for the Abramowitz-Stegun form above on
// packed fp32 (half rate on gfx950): the GELU arithmetic was 58 of the 319 us of an mlp.fc1 launch at B = 36
// (profiles/r04_epilogue_table.txt: no-store build vs K-loop-only build).  Past |x| = 6.22 (|z| = 4.4) h is held at 3e-10: for very negative x
// the result is -3e-10 |x| instead of -> 0 (2e-5 at the fp16 limit, below one fp16 ulp of anything the next GEMM adds it to), and +-inf gives
// NaN like every other arithmetic on a non-finite accumulator -- lseg_check_range reports those; a compare + select per element to return
// max(x, 0) there would cost the issue-bound epilogue 2 more VALU instructions per element.
__device__ __forceinline__ float gelu_fast(float x) {
    const float u = fminf(fabsf(x), 6.222539901733398f);
    float r = 2.1934425603831187e-05f;
    r = fmaf(r, u, -0.000676676572766155f);
    r = fmaf(r, u, 0.007795793004333973f);
    r = fmaf(r, u, -0.0530061200261116f);
    r = fmaf(r, u, -0.45904305577278137f);
    r = fmaf(r, u, -1.151124119758606f);
    const float h = __builtin_amdgcn_exp2f(fmaf(r, u, -1.0f));
    return fmaf(-fabsf(x), h, fmaxf(x, 0.f));
}

// ReLU on 8 packed 16-bit floats (bf16 or fp16): negative <=> sign bit <=> negative int16.
__device__ __forceinline__ i32x4_t relu_frag(i32x4_t v) {
    i32x4_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = v[i];
        const int lo = (x & 0x8000) ? 0 : (x & 0xffff);
        const int hi = (x < 0) ? 0 : (x & 0xffff0000);
        r[i] = lo | hi;
    }
    return r;
}

// ---- epilogue addressing: every output map is separable, off(m, n) = rowpart(m) + colpart(n) --------
struct ColPart { size_t off; int which; };   // which: 0 -> C, 1 -> Ck, 2 -> Cv (MAP_QKV only)

__device__ __forceinline__ ColPart col_part(const GemmArgs& g, int n) {
    ColPart c{(size_t)n, 0};
    switch (g.map_mode) {
        case MAP_PIXSHUF: {
            const int ij = n / g.ps_C, co = n - ij * g.ps_C;
            const int i = ij / g.ps_s, j = ij - i * g.ps_s;
            const int Wd = g.wo * g.ps_s + 2;
            c.off = ((size_t)i * Wd + j) * g.ps_C + co;
        } break;
        case MAP_QKV: {
            const int which = n / g.qkv_dim, rem = n - which * g.qkv_dim;
            const int head = rem >> 6, d = rem & 63;
            c.which = which;
            c.off = which == 2 ? ((size_t)head * 64 + d) * g.qkv_npad : (size_t)head * g.qkv_npad * 64 + d;
        } break;
        case MAP_NCHW: c.off = (size_t)n * g.p_div; break;
        case MAP_LABELPLANES: {
            const int b = n / g.p_div;
            c.off = (size_t)b * g.M * g.p_div + (size_t)(n - b * g.p_div);
        } break;
        default: break;
    }
    return c;
}
// rowpart for the q/k (or only) layout in .x, for the transposed-v layout in .y (MAP_QKV)
__device__ __forceinline__ void row_part(const GemmArgs& g, int m, size_t& r0, size_t& r1) {
    r1 = 0;
    switch (g.map_mode) {
        case MAP_LINEAR: r0 = (size_t)m * g.ldc; break;
        case MAP_PERIODIC: {
            const int q = m / g.p_div, r = m - q * g.p_div;
            r0 = (size_t)(q * g.p_mul + r + g.p_off) * g.ldc;
        } break;
        case MAP_PADDED: {
            const int hw = g.ho * g.wo;
            const int b = m / hw, p = m - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            r0 = ((size_t)(b * (g.ho + 2) + y + 1) * (g.wo + 2) + x + 1) * g.ldc;
        } break;
        case MAP_PIXSHUF: {
            const int hw = g.ho * g.wo;
            const int b = m / hw, p = m - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            const int Hd = g.ho * g.ps_s + 2, Wd = g.wo * g.ps_s + 2;
            r0 = ((size_t)(b * Hd + y * g.ps_s + 1) * Wd + x * g.ps_s + 1) * g.ps_C;
        } break;
        case MAP_QKV: {
            const int b = m / g.qkv_ntok, t = m - b * g.qkv_ntok;
            r0 = ((size_t)b * g.qkv_heads * g.qkv_npad + t) * 64;
            r1 = (size_t)b * g.qkv_heads * 64 * g.qkv_npad + t;
        } break;
        case MAP_NCHW: {
            const int b = m / g.p_div, p = m - b * g.p_div;
            r0 = (size_t)b * g.N * g.p_div + p;
        } break;
        case MAP_LABELPLANES: r0 = (size_t)m * g.p_div; break;
        default: r0 = 0; break;
    }
}

__device__ __forceinline__ float4 load4_as_f32(const void* p, size_t off, int dtype) {
    if (dtype == DT_F32) return *reinterpret_cast<const float4*>((const float*)p + off);
    const uint2 u = *reinterpret_cast<const uint2*>((const uint16_t*)p + off);
    float4 r;
    if (dtype == DT_F16) {
        r.x = f16_to_f32((uint16_t)u.x); r.y = f16_to_f32((uint16_t)(u.x >> 16));
        r.z = f16_to_f32((uint16_t)u.y); r.w = f16_to_f32((uint16_t)(u.y >> 16));
    } else {
        r.x = bf16_to_f32((uint16_t)u.x); r.y = bf16_to_f32((uint16_t)(u.x >> 16));
        r.z = bf16_to_f32((uint16_t)u.y); r.w = bf16_to_f32((uint16_t)(u.y >> 16));
    }
    return r;
}

template <typename T>
__device__ __forceinline__ void apply_act(const GemmArgs& g, float (&v)[4]) {
    if (g.round_mid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = to_f32<T>(from_f32<T>(v[r]));
    }
    if (g.act == ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_fast(v[r]);
    } else if (g.act == ACT_QUICKGELU) {
        // x * sigmoid(1.702 x); with round_mid every step is rounded like the fp16 eager ops
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = 1.702f * v[r];
            if (g.round_mid) t = to_f32<T>(from_f32<T>(t));
            float s = 1.0f / (1.0f + __expf(-t));
            if (g.round_mid) s = to_f32<T>(from_f32<T>(s));
            v[r] = v[r] * s;
            if (g.round_mid) v[r] = to_f32<T>(from_f32<T>(v[r]));
        }
    } else if (g.act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
}

// (hi, lo) fp16 pair of a value: hi = fp16(v), lo = fp16(v - hi)
__device__ __forceinline__ void store_split(void* dst, size_t off, size_t plane, float v) {
    const float hi = round_f16(v);
    ((uint16_t*)dst)[off] = f32_to_f16(hi);
    ((uint16_t*)dst)[off + plane] = f32_to_f16(v - hi);
}

__device__ __forceinline__ void store4(void* dst, size_t off, int dtype, const float (&v)[4]) {
    if (dtype == DT_F32) {
        *reinterpret_cast<float4*>((float*)dst + off) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 pk;
        pk.x = pack2_dt(v[0], v[1], dtype);
        pk.y = pack2_dt(v[2], v[3], dtype);
        *reinterpret_cast<uint2*>((uint16_t*)dst + off) = pk;
    }
}

// Per-tile column info of this lane: first column of each 4-wide group, its address part, its bias.
// `wide` = the permlane-widened 16-byte store path applies to this launch (16-bit output, unit-stride
// map, every 32-column pair fully in range); cpw[] = address part of the 8 columns a lane owns then.
template <int NI>
__device__ __forceinline__ bool epilogue_cols(const GemmArgs& g, int n_first, int lane, int (&ncol)[NI], ColPart (&cp)[NI],
                                              ColPart (&cpw)[NI / 2], float4 (&bias)[NI]) {
    const bool wide = g.out_dtype != DT_F32 && (g.N & 31) == 0 && g.map_mode != MAP_NCHW && (g.dbg & 3) == 0 && !g.split &&
                      (g.map_mode != MAP_PIXSHUF || (g.ps_C & 31) == 0);
    {
        const int r16 = lane >> 4;
        const int pair0 = n_first - r16 * 4;                  // first column of this lane's first sub-tile pair
#pragma unroll
        for (int p = 0; p < NI / 2; ++p) {
            const int c = pair0 + p * 32 + (r16 & 1) * 16 + (r16 >> 1) * 8;
            cpw[p] = col_part(g, c < g.N ? c : 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n_first + i * 16;
        ncol[i] = n;
        cp[i] = col_part(g, n < g.N ? n : 0);
        bias[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias && n < g.N) {
            const int bn = g.bias_mod ? (n % g.bias_mod) : n;
            if ((g.N & 3) == 0) {
                bias[i] = *reinterpret_cast<const float4*>(g.bias + bn);
            } else {
                bias[i].x = g.bias[bn];
                if (n + 1 < g.N) bias[i].y = g.bias[bn + 1];
                if (n + 2 < g.N) bias[i].z = g.bias[bn + 2];
                if (n + 3 < g.N) bias[i].w = g.bias[bn + 3];
            }
        }
    }
    return wide;
}

// One output row m of this lane: NI groups of 4 consecutive columns.  All residual loads of the
// row are issued before the first store (the residual may alias C: x += f(x) in place).
template <typename T, int NI>
__device__ __forceinline__ void epilogue_row(const GemmArgs& g, int m, const int (&ncol)[NI], const ColPart (&cp)[NI],
                                             const ColPart (&cpw)[NI / 2], bool wide, const float4 (&bias)[NI],
                                             f32x4_t (&acc)[NI], size_t c_extra) {
    size_t r0, r1;
    row_part(g, m, r0, r1);
    r0 += c_extra;                                   // split-K: this work item's partial-result slab
    size_t rres = 0;
    if (g.res_mode == RES_PERIODIC) rres = (size_t)((m % g.p_div) + g.p_off) * g.ldr;
    const bool fast = (g.N % 4) == 0;          // every group of 4 columns is fully in range
    float v[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i][0] = acc[i][0] + bias[i].x; v[i][1] = acc[i][1] + bias[i].y;
        v[i][2] = acc[i][2] + bias[i].z; v[i][3] = acc[i][3] + bias[i].w;
    }
    if (g.res_mode == RES_NONE) {
#pragma unroll
        for (int i = 0; i < NI; ++i) apply_act<T>(g, v[i]);
    } else {
        float4 rv[NI], rv2[NI];
        const bool strided = g.map_mode == MAP_NCHW || g.map_mode == MAP_QKV;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            rv[i] = make_float4(0.f, 0.f, 0.f, 0.f); rv2[i] = rv[i];
            if (ncol[i] >= g.N) continue;
            const size_t roff = g.res_mode == RES_PERIODIC ? rres + ncol[i] : r0 + cp[i].off;
            if (fast && !strided) {
                rv[i] = load4_as_f32(g.res, roff, g.res_dtype);
                if (g.res2) rv2[i] = load4_as_f32(g.res2, roff, g.res_dtype);
                if (g.split && g.res_dtype != DT_F32) {          // 16-bit residual maps are (hi, lo) planes
                    const float4 l1 = load4_as_f32(g.res, roff + g.res_plane, g.res_dtype);
                    rv[i].x += l1.x; rv[i].y += l1.y; rv[i].z += l1.z; rv[i].w += l1.w;
                    if (g.res2) {
                        const float4 l2 = load4_as_f32(g.res2, roff + g.res_plane, g.res_dtype);
                        rv2[i].x += l2.x; rv2[i].y += l2.y; rv2[i].z += l2.z; rv2[i].w += l2.w;
                    }
                }
            } else {
                float t[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < 4; ++r)
                    if (ncol[i] + r < g.N) t[r] = load_as_f32(g.res, roff + r, g.res_dtype) + (g.res2 ? load_as_f32(g.res2, roff + r, g.res_dtype) : 0.f);
                rv[i] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            apply_act<T>(g, v[i]);
            v[i][0] += rv[i].x + rv2[i].x; v[i][1] += rv[i].y + rv2[i].y;
            v[i][2] += rv[i].z + rv2[i].z; v[i][3] += rv[i].w + rv2[i].w;
        }
    }
    if ((g.dbg & 3) == 1) {   // ablation: keep the values live, skip the stores
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(v[i][0]), "v"(v[i][1]), "v"(v[i][2]), "v"(v[i][3]));
        return;
    }
    // 16-bit outputs: a lane holds 8 bytes (4 columns) per 16-column sub-tile.  v_permlane16_swap on
    // a pair of sub-tiles regroups them so every lane owns 16 contiguous bytes and one store
    // instruction writes 16 rows x 64 contiguous bytes (half the store instructions; T21-style):
    //   lane row r16 = lane>>4 ends up with columns  pair_base + (r16&1)*16 + (r16>>1)*8 .. +7
    if (wide) {
#pragma unroll
        for (int i = 0; i < NI; i += 2) {
            if (ncol[i] >= g.N) continue;      // whole 32-column pair beyond N (wave-uniform: N % 32 == 0), e.g. N = 64 on a 128-wide tile
            if (cpw[i / 2].which == 2) {       // V^T of MAP_QKV: strided, scalar stores
#pragma unroll
                for (int q = i; q < i + 2; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        store_from_f32(g.Cv, r1 + cp[q].off + (size_t)r * g.qkv_npad, g.out_dtype, v[q][r]);
                continue;
            }
            const uint32_t a0 = pack2_dt(v[i][0], v[i][1], g.out_dtype), a1 = pack2_dt(v[i][2], v[i][3], g.out_dtype);
            const uint32_t b0 = pack2_dt(v[i + 1][0], v[i + 1][1], g.out_dtype), b1 = pack2_dt(v[i + 1][2], v[i + 1][3], g.out_dtype);
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            void* dst = cpw[i / 2].which == 1 ? g.Ck : g.C;
            *reinterpret_cast<uint4*>((uint16_t*)dst + r0 + cpw[i / 2].off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (ncol[i] >= g.N) continue;
        void* dst = g.C;
        size_t off = r0 + cp[i].off, estride = 1;
        if (g.map_mode == MAP_QKV) {
            if (cp[i].which == 2) { dst = g.Cv; off = r1 + cp[i].off; estride = g.qkv_npad; }
            else if (cp[i].which == 1) dst = g.Ck;
        } else if (g.map_mode == MAP_NCHW) {
            estride = g.p_div;
        }
        if (g.split && g.out_dtype != DT_F32) {
            const size_t plane = dst == g.Cv && g.map_mode == MAP_QKV ? g.cv_plane : (dst == g.Ck && g.map_mode == MAP_QKV ? g.ck_plane : g.c_plane);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ncol[i] + r < g.N) store_split(dst, off + r * estride, plane, v[i][r]);
        } else if (estride == 1 && fast) {
            store4(dst, off, g.out_dtype, v[i]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ncol[i] + r < g.N) store_from_f32(dst, off + r * estride, g.out_dtype, v[i][r]);
        }
    }
}

// ---- specialised epilogues of the ViT's four GEMMs ------------------------------------------------
// The generic epilogue above decodes map/residual/dtype/activation at run time; measured with
// tools/gemm_phase_probe.py it cost ~9,500 cycles per 128x128 tile (a quarter of a K=1024 tile: ~3,000
// waiting for the bias loads, ~3,700 in run-time-dispatched conversion code, ~2,900 in stores).  The hot
// shapes get compile-time variants: bias prefetched at the start of the tile, straight-line math,
// permlane-widened 16-byte stores.  Requirements (checked by select_epi): N % 128 == 0, a bias,
// no round_mid, 16-byte aligned rows.
enum Epi {
    EPI_GENERIC = 0,
    EPI_LIN16 = 1,        // C[m*ldc+n] = T(acc + bias)                      (attention proj inputs, readout, ...)
    EPI_LIN16_GELU = 2,   // C[m*ldc+n] = T(gelu(acc + bias))                (MLP fc1)
    EPI_RES32 = 3,        // C[m*ldc+n] = res[m*ldc+n] + acc + bias, fp32    (attention proj / MLP fc2 on the residual stream)
    EPI_QKV16 = 4,        // MAP_QKV, q,k -> [b,head,t,d], v -> [b,head,d,t]
    EPI_PAD16 = 5,        // MAP_PADDED NHWC (1-pixel zero border), T(act(acc + bias) [+ res [+ res2]]): the DPT head's 3x3 convs
    EPI_LIN16_F16 = 6,    // MAP_LINEAR, C = fp16(acc + bias) from bf16 operands: the commuted head's g (the fp16 operand of the correlation)
    EPI_PIX16 = 7,        // MAP_PIXSHUF, T(acc + bias[n % C]): ConvTranspose2d(k = s) as a GEMM whose columns scatter to the s x s sub-pixels
    EPI_LIN16_GELU2 = 9,  // C_pre[m*ldc+n] = z = T(acc + bias), C[m*ldc+n] = T(gelu(z))   (MLP fc1 of the training forward)
    EPI_LIN16_DGELU = 10, // C[m*ldc+n] = T((acc + bias) * gelu'(pre[m*ldc+n]))            (dX of the MLP fc2, through the GELU)
    EPI_LIN32 = 11,       // C[m*ldc+n] = acc + bias, fp32                       (training: head1's output feeds the fp32 L2 normalisation)
    EPI_PART32 = 8,       // split-K partial slab: C[split * c_split_stride + m*ldc + n] = acc, fp32, no bias (weight gradients; the residual GEMMs
                          // of small batches, whose slabs the following LayerNorm sums into the fp32 residual stream)
};

// Epilogue stores of the specialised epilogues go through these three helpers so that the attribution builds (tools/epilogue_table.py,
// make VARIANT=...: -DGEMM_EPI_ABL=1 keeps the loads and the arithmetic and drops the stores; 2 drops the whole epilogue, 3 drops the
// residual loads of EPI_RES32, 4 writes V like K in EPI_QKV16) measure the K-loop and each epilogue part of the SAME kernel.
#ifndef GEMM_EPI_ABL
#define GEMM_EPI_ABL 0
#endif
// GEMM_NT (A/B builds, tools): 1 = the specialised epilogues' stores are NON-TEMPORAL (the output streams past L2 instead of evicting the operand
// panels the XCD's other tiles are about to re-read), 2 = the residual loads of EPI_RES32 as well
#ifndef GEMM_NT
#define GEMM_NT 0
#endif
typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_u4(void* p, const uint4 v) {
    if constexpr (GEMM_EPI_ABL == 1) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(p));
    else if constexpr (GEMM_NT >= 1) __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(p));
    else *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void st_f4(float* p, const float4 v) {
    if constexpr (GEMM_EPI_ABL == 1) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(p));
    else if constexpr (GEMM_NT >= 1) __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(p));
    else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ void st_h(uint16_t* p, const uint16_t v) {
    if constexpr (GEMM_EPI_ABL == 1) asm volatile("" ::"v"(v), "v"(p));
    else *p = v;
}
__device__ __forceinline__ float4 ld_res4(const float* p) {
    if constexpr (GEMM_NT >= 2) {
        const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}

// two adjacent 16-column sub-tiles (4 columns per lane each) -> 8 contiguous columns (16 bytes) per lane
template <typename T>
__device__ __forceinline__ uint4 widen16(const float (&x)[4], const float (&y)[4]) {
    const uint32_t a0 = pack2<T>(x[0], x[1]), a1 = pack2<T>(x[2], x[3]);
    const uint32_t b0 = pack2<T>(y[0], y[1]), b1 = pack2<T>(y[2], y[3]);
    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    return make_uint4(s0[0], s1[0], s0[1], s1[1]);
}

// GEMM_RES_PRELOAD (round 6): EPI_RES32 with the RESIDUAL AS THE ACCUMULATORS' START VALUE.  The epilogue of round 5 loaded the lane's residual
// sub-tile AFTER the K-loop (the accumulators own the registers it would land in) -- 133 MB per tile round that every CU requests at once and
// then waits for: the exposed half of the read-modify-write (profiles/r04_epilogue_table.txt).  Here the accumulators of tile i+1 are loaded with
// its residual by the epilogue of tile i, chunk by chunk right behind the stores that free them: stores and loads are both posted, nothing in
// the epilogue waits for memory, and the loads land under the epilogue's remaining chunks and the next tile's first fragment reads (the first
// MFMA that touches a chunk waits for it).  C = (res + A W^T) + bias: the products accumulate ON the residual (fp32, ~1e-6 |res| of rounding)
// instead of being added to it last -- the same association as the hand-scheduled kernel (gemm_asm.hip).
// MEASURED (profiles/r06_gemm_asm.txt, same box, interleaved): attn.proj 93.5-94.3 vs 93.6-98.7 us, mlp.fc2 258-261 vs 256-261 us -- NO gain: the
// read-modify-write of a tile costs a CU ~25 us whether the loads are waited for in the epilogue or at the next tile's first MFMA, on the whole
// chip or on half of it (r04_epilogue_table.txt "partial") -- it is bound by what one CU can keep in flight against HBM latency (~20 GB/s per
// CU), not by the wait structure and not by the chip's bandwidth.  Kept as a build switch, OFF (round-5 association of the sums).
#ifndef GEMM_RES_PRELOAD
#define GEMM_RES_PRELOAD 0
#endif
template <int MI, int NI>
__device__ __forceinline__ void res_preload(const GemmArgs& g, f32x4_t (&acc)[NI][MI], int mrow0, int ncol0, int lane) {
    const int r16 = lane >> 4, ml = lane & 15;
    const float* rbase = (const float*)g.res + ncol0 + r16 * 4;
    static_for<0, MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        int m = mrow0 + j * 16 + ml;
        if (m > g.M - 1) m = g.M - 1;
        const float* rp = rbase + (size_t)m * g.ldc;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(rp + i * 16);
            acc[i][j] = f32x4_t{v.x, v.y, v.z, v.w};
        }
    });
}
// the epilogue of that form: store (acc + bias) chunk by chunk; `next` = a further tile of this workgroup follows, whose residual (sub-tile at
// nrow0 / ncol_n) goes into the registers each chunk's stores have just read
template <int MI, int NI>
__device__ __forceinline__ void res_store_and_preload(const GemmArgs& g, f32x4_t (&acc)[NI][MI], int mrow0, int ncol0, int lane, const float4 (&bias)[NI],
                                                      bool next, int nrow0, int ncol_n) {
    const int r16 = lane >> 4, ml = lane & 15;
    float* cbase = (float*)g.C + ncol0 + r16 * 4;
    const float* rbase = (const float*)g.res + ncol_n + r16 * 4;
    static_for<0, MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m = mrow0 + j * 16 + ml;
        float* p = cbase + (size_t)m * g.ldc;
        if (m < g.M) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                st_f4(p + i * 16, make_float4(acc[i][j][0] + bias[i].x, acc[i][j][1] + bias[i].y, acc[i][j][2] + bias[i].z, acc[i][j][3] + bias[i].w));
        }
        if (next) {                               // wave-uniform
            int mn = nrow0 + j * 16 + ml;
            if (mn > g.M - 1) mn = g.M - 1;
            const float* rp = rbase + (size_t)mn * g.ldc;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(rp + i * 16);
                acc[i][j] = f32x4_t{v.x, v.y, v.z, v.w};
            }
        }
    });
}

// mrow0 / ncol0: first row / column of this WAVE's sub-tile (wave-uniform).  acc[i][j]: rows
// mrow0 + j*16 + (lane&15), columns ncol0 + i*16 + (lane>>4)*4 .. +3.
template <typename T, int EPI, int MI, int NI>
__device__ __forceinline__ void fast_epilogue(const GemmArgs& g, const GemmArgs& gk, f32x4_t (&acc)[NI][MI], int mrow0, int ncol0, int lane,
                                              const float4 (&bias)[NI], size_t c_extra) {
    const int r16 = lane >> 4, ml = lane & 15;
    const int cw = (r16 & 1) * 16 + (r16 >> 1) * 8;       // lane's 8 columns inside a 32-column pair after the swap
    auto biased = [&](auto ic, auto jc, float (&v)[4]) {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
        v[0] = acc[i][j][0] + bias[i].x; v[1] = acc[i][j][1] + bias[i].y;
        v[2] = acc[i][j][2] + bias[i].z; v[3] = acc[i][j][3] + bias[i].w;
    };
    if constexpr (EPI == EPI_LIN16 || EPI == EPI_LIN16_GELU || EPI == EPI_LIN16_F16 || EPI == EPI_LIN16_GELU2 || EPI == EPI_LIN16_DGELU) {
        using OT = typename std::conditional<EPI == EPI_LIN16_F16, F16, T>::type;
        uint16_t* cbase = (uint16_t*)g.C + ncol0 + cw;
        static_for<0, MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int m = mrow0 + j * 16 + ml;
            uint16_t* p = cbase + (size_t)m * g.ldc;
            uint2 zv[EPI == EPI_LIN16_DGELU ? NI : 1];
            if constexpr (EPI == EPI_LIN16_DGELU) {      // this lane's pre-activations of the row: all loads in flight before the math
                const uint16_t* zp = (const uint16_t*)g.dgelu_pre + (size_t)(m < g.M ? m : g.M - 1) * g.ldc + ncol0 + r16 * 4;
#pragma unroll
                for (int i = 0; i < NI; ++i) zv[i] = *reinterpret_cast<const uint2*>(zp + i * 16);
            }
            static_for<0, NI / 2>([&](auto pc) {
                constexpr int i = 2 * decltype(pc)::value;
                float x[4], y[4];
                biased(std::integral_constant<int, i>{}, jc, x);
                biased(std::integral_constant<int, i + 1>{}, jc, y);
                if constexpr (EPI == EPI_LIN16_DGELU) {
                    const uint2 u = zv[i], v = zv[i + 1];
                    x[0] *= gelu_grad_as(to_f32<T>((uint16_t)u.x)); x[1] *= gelu_grad_as(to_f32<T>((uint16_t)(u.x >> 16)));
                    x[2] *= gelu_grad_as(to_f32<T>((uint16_t)u.y)); x[3] *= gelu_grad_as(to_f32<T>((uint16_t)(u.y >> 16)));
                    y[0] *= gelu_grad_as(to_f32<T>((uint16_t)v.x)); y[1] *= gelu_grad_as(to_f32<T>((uint16_t)(v.x >> 16)));
                    y[2] *= gelu_grad_as(to_f32<T>((uint16_t)v.y)); y[3] *= gelu_grad_as(to_f32<T>((uint16_t)(v.y >> 16)));
                }
                if constexpr (EPI == EPI_LIN16_GELU2) {
                    // store the rounded pre-activation, then take the GELU of exactly those values
                    const uint32_t a0 = pack2<T>(x[0], x[1]), a1 = pack2<T>(x[2], x[3]), b0 = pack2<T>(y[0], y[1]), b1 = pack2<T>(y[2], y[3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    if (m < g.M) *reinterpret_cast<uint4*>((uint16_t*)g.C_pre + (size_t)m * g.ldc + ncol0 + cw + i * 16) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    x[0] = gelu_fast(to_f32<T>((uint16_t)a0)); x[1] = gelu_fast(to_f32<T>((uint16_t)(a0 >> 16)));
                    x[2] = gelu_fast(to_f32<T>((uint16_t)a1)); x[3] = gelu_fast(to_f32<T>((uint16_t)(a1 >> 16)));
                    y[0] = gelu_fast(to_f32<T>((uint16_t)b0)); y[1] = gelu_fast(to_f32<T>((uint16_t)(b0 >> 16)));
                    y[2] = gelu_fast(to_f32<T>((uint16_t)b1)); y[3] = gelu_fast(to_f32<T>((uint16_t)(b1 >> 16)));
                }
                if constexpr (EPI == EPI_LIN16_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { x[r] = gelu_fast(x[r]); y[r] = gelu_fast(y[r]); }
                }
                const uint4 o = widen16<OT>(x, y);
                if (m < g.M) st_u4(p + i * 16, o);
            });
        });
    } else if constexpr (EPI == EPI_RES32) {
        // every residual load of the sub-tile is in flight before the first store (in place: x += f(x))
        float* cbase = (float*)g.C + ncol0 + r16 * 4;
        const float* rbase = (const float*)g.res + ncol0 + r16 * 4;
        // rows per chunk: <= 32 registers of prefetched residual.  With 64 the 256x256 instance spilled 156 bytes, and the reloads sat in
        // the K-loop's tile-advance paths behind s_waitcnt vmcnt(0) -- two drains of the DMA ring per tile.  A/B in the engine (B = 36, fp16,
        // lease F): attn.proj 124.4 -> 108.3 us, mlp.fc2 279.4 -> 261.3 us per launch.
        // (the 64-accumulator tiles keep 64: they have the registers, and a bigger batch of loads per wait)
        constexpr int JC0 = (MI * NI > 16 ? 32 : 64) / 4 / NI;
        constexpr int JC = JC0 < 1 ? 1 : (JC0 > MI ? MI : JC0);
        static_for<0, MI / JC>([&](auto cc) {
            constexpr int jb = decltype(cc)::value * JC;
            float4 rv[JC][NI];
            static_for<0, JC>([&](auto jc) {
                constexpr int j = jb + decltype(jc)::value;
                int m = mrow0 + j * 16 + ml;
                if (m > g.M - 1) m = g.M - 1;
                const float* rp = rbase + (size_t)m * g.ldc;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if constexpr (GEMM_EPI_ABL == 3) rv[decltype(jc)::value][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    else rv[decltype(jc)::value][i] = ld_res4(rp + i * 16);
                }
            });
            static_for<0, JC>([&](auto jc) {
                constexpr int jl = decltype(jc)::value, j = jb + jl;
                const int m = mrow0 + j * 16 + ml;
                float* p = cbase + (size_t)m * g.ldc;
                static_for<0, NI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    float v[4];
                    biased(ic, std::integral_constant<int, j>{}, v);
                    if (m < g.M)
                        st_f4(p + i * 16, make_float4(v[0] + rv[jl][i].x, v[1] + rv[jl][i].y, v[2] + rv[jl][i].z, v[3] + rv[jl][i].w));
                });
            });
        });
    } else if constexpr (EPI == EPI_LIN32) {
        float* cbase = (float*)g.C + ncol0 + r16 * 4;
        static_for<0, MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int m = mrow0 + j * 16 + ml;
            float* p = cbase + (size_t)m * g.ldc;
            static_for<0, NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float v[4];
                biased(ic, jc, v);
                if (m < g.M) *reinterpret_cast<float4*>(p + i * 16) = make_float4(v[0], v[1], v[2], v[3]);
            });
        });
    } else if constexpr (EPI == EPI_PART32) {
        float* cbase = (float*)g.C + c_extra + ncol0 + r16 * 4;
        static_for<0, MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int m = mrow0 + j * 16 + ml;
            float* p = cbase + (size_t)m * g.ldc;
            static_for<0, NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (m < g.M) *reinterpret_cast<float4*>(p + i * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            });
        });
    } else if constexpr (EPI == EPI_PAD16) {
        // m = (b, y, x) on the (ho, wo) output grid -> row of the padded NHWC map; columns are contiguous channels.
        // Residuals (the RCU skip input and, for the fusion add, the other path) live at the destination offsets.
        const bool relu = g.act == ACT_RELU, has_res = g.res_mode == RES_DEST, has_res2 = has_res && g.res2 != nullptr;
        const int hw = g.ho * g.wo;
        static_for<0, MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int m = mrow0 + j * 16 + ml;
            const int mm = m < g.M ? m : g.M - 1;
            const int b = mm / hw, p = mm - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            const size_t r0 = ((size_t)(b * (g.ho + 2) + y + 1) * (g.wo + 2) + x + 1) * g.ldc + ncol0;
            uint2 rv[NI], rv2[NI];
            if (has_res) {
#pragma unroll
                for (int i = 0; i < NI; ++i) rv[i] = *reinterpret_cast<const uint2*>((const uint16_t*)g.res + r0 + i * 16 + r16 * 4);
                if (has_res2) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) rv2[i] = *reinterpret_cast<const uint2*>((const uint16_t*)g.res2 + r0 + i * 16 + r16 * 4);
                }
            }
            uint16_t* prow = (uint16_t*)g.C + r0 + cw;
            static_for<0, NI / 2>([&](auto pc) {
                constexpr int i = 2 * decltype(pc)::value;
                float x4[2][4];
                biased(std::integral_constant<int, i>{}, jc, x4[0]);
                biased(std::integral_constant<int, i + 1>{}, jc, x4[1]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) x4[q][r] = x4[q][r] > 0.f ? x4[q][r] : 0.f;
                    }
                    if (has_res) {
                        const uint2 u = rv[i + q];
                        x4[q][0] += to_f32<T>((uint16_t)u.x); x4[q][1] += to_f32<T>((uint16_t)(u.x >> 16));
                        x4[q][2] += to_f32<T>((uint16_t)u.y); x4[q][3] += to_f32<T>((uint16_t)(u.y >> 16));
                        if (has_res2) {
                            const uint2 u2 = rv2[i + q];
                            x4[q][0] += to_f32<T>((uint16_t)u2.x); x4[q][1] += to_f32<T>((uint16_t)(u2.x >> 16));
                            x4[q][2] += to_f32<T>((uint16_t)u2.y); x4[q][3] += to_f32<T>((uint16_t)(u2.y >> 16));
                        }
                    }
                }
                const uint4 o = widen16<T>(x4[0], x4[1]);
                if (m < g.M) {
                    *reinterpret_cast<uint4*>(prow + i * 16) = o;
                    if (g.C_relu) {          // second copy with the negative halves cleared (sign bit of a bf16 / fp16 half)
                        auto rl = [](uint32_t w) { return ((w & 0x8000u) ? 0u : (w & 0xffffu)) | ((w & 0x80000000u) ? 0u : (w & 0xffff0000u)); };
                        *reinterpret_cast<uint4*>((uint16_t*)g.C_relu + r0 + cw + i * 16) = make_uint4(rl(o.x), rl(o.y), rl(o.z), rl(o.w));
                    }
                }
            });
        });
    } else if constexpr (EPI == EPI_PIX16) {
        // m = (b, y, x) on the (ho, wo) grid -> pixel (y*s + i, x*s + j) of the padded [B, ho*s + 2, wo*s + 2, C] map for column
        // n = (i*s + j)*C + co; a 32-column pair lies inside one (i, j) group (C % 32 == 0): 16-byte stores of 8 channels
        const int hw = g.ho * g.wo, Hd = g.ho * g.ps_s + 2, Wd = g.wo * g.ps_s + 2;
        static_for<0, MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int m = mrow0 + j * 16 + ml;
            const int mm = m < g.M ? m : g.M - 1;
            const int b = mm / hw, p = mm - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            const size_t r0 = ((size_t)(b * Hd + y * g.ps_s + 1) * Wd + x * g.ps_s + 1) * g.ps_C;
            static_for<0, NI / 2>([&](auto pc) {
                constexpr int i = 2 * decltype(pc)::value;
                const int c0 = ncol0 + i * 16;
                const int ij = c0 / g.ps_C, co = c0 - ij * g.ps_C;
                const int si = ij / g.ps_s, sj = ij - si * g.ps_s;
                float xx[4], yy[4];
                biased(std::integral_constant<int, i>{}, jc, xx);
                biased(std::integral_constant<int, i + 1>{}, jc, yy);
                const uint4 o = widen16<T>(xx, yy);
                if (m < g.M) *reinterpret_cast<uint4*>((uint16_t*)g.C + r0 + ((size_t)si * Wd + sj) * g.ps_C + co + cw) = o;
            });
        });
    } else if constexpr (EPI == EPI_QKV16) {
        // Column side on the scalar unit (ncol0 is wave-uniform): every 32-column pair of a wave lies inside one head of one of q / k / v,
        // and -- select_epi checks qkv_dim % 128 == 0 -- ALL columns of a wave inside one of the three: ONE division per tile decides the
        // destination, the per-pair offsets are shifts of rem0 + 32 * pair.  32-bit ELEMENT offsets: q, k [b, head, t, d]: head * npad * 64 + d0;
        // v^T [b, head, d, t]: (head * 64 + d0) * npad.  Row side: (b, t) of the wave's first row by one division, the lane's rows by adds and
        // ONE select when an image has at least as many tokens as a wave has rows (ViT: 901), a per-row division otherwise (the text tower's
        // <= 77 positions).  No loop: a `while (t >= ntok)` here made hipcc build exec-masked loops whose preheaders carry s_waitcnt vmcnt(0)
        // -- four full drains of the epilogue's stores and of the next tile's DMA per tile (qkv 197 -> 223 us in the engine at B = 36).
        // (Round 3 decoded which / head per pair and b / t per row with run-time integer divisions inside the unrolled code: ~370 v_readlane
        // SGPR reloads and 221 64-bit address adds per wave tile, 24 us of a 252 us launch before a single store --
        // profiles/r04_epilogue_table.txt.)
        const int c0 = __builtin_amdgcn_readfirstlane(ncol0);
        const int which = c0 / gk.qkv_dim, rem0 = c0 - which * gk.qkv_dim;
        const int ntok = gk.qkv_ntok, npad = gk.qkv_npad, Mr = gk.M;
        const int mr0 = __builtin_amdgcn_readfirstlane(mrow0 < Mr ? mrow0 : Mr - 1);
        const int b0 = mr0 / ntok, t0 = mr0 - b0 * ntok;
        const uint32_t hn = (uint32_t)gk.qkv_heads * (uint32_t)npad;
        auto row_bt = [&](int j, int& b, int& t) {            // (image, token) of this lane's row j * 16 + ml of the wave tile
            if (ntok >= MI * 16) {
                t = t0 + j * 16 + ml;
                const bool wrap = t >= ntok;
                t -= wrap ? ntok : 0;
                b = b0 + (wrap ? 1 : 0);
            } else {
                int mm = mrow0 + j * 16 + ml;
                mm = mm < Mr ? mm : Mr - 1;
                b = mm / ntok;
                t = mm - b * ntok;
            }
        };
        if (which < 2 || GEMM_EPI_ABL == 4) {
            uint16_t* base = (uint16_t*)(which == 2 ? gk.Cv : which ? gk.Ck : gk.C);         // wave-uniform: SGPR base + 32-bit lane offset
            const float qs = which == 0 && gk.qkv_qscale != 0.f ? gk.qkv_qscale : 1.0f;      // q carries the softmax scale
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                int b, t;
                row_bt(j, b, t);
                const uint32_t rq = ((uint32_t)b * hn + (uint32_t)t) * 64u + (uint32_t)cw;
                const bool ok = mrow0 + j * 16 + ml < Mr;
                static_for<0, NI / 2>([&](auto pc) {
                    constexpr int pi = decltype(pc)::value, i = 2 * pi;
                    const uint32_t rem = (uint32_t)rem0 + 32u * pi;
                    const uint32_t cb = (rem >> 6) * (uint32_t)npad * 64u + (rem & 63u);
                    float x[4], y[4];
                    biased(std::integral_constant<int, i>{}, jc, x);
                    biased(std::integral_constant<int, i + 1>{}, jc, y);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { x[r] *= qs; y[r] *= qs; }
                    const uint4 o = widen16<T>(x, y);
                    if (ok) st_u4(reinterpret_cast<char*>(base) + ((rq + cb) << 1), o);       // 32-bit BYTE offset on a uniform base (elements < 2^31)
                });
            });
        } else {
            // V^T [b, head, d, t]: t is the contiguous axis; 16 lanes write 16 consecutive tokens, 2 bytes each
            uint16_t* base = (uint16_t*)gk.Cv;
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                int b, t;
                row_bt(j, b, t);
                const uint32_t rv = (uint32_t)b * hn * 64u + (uint32_t)t + (uint32_t)(r16 * 4) * (uint32_t)npad;
                const bool ok = mrow0 + j * 16 + ml < Mr;
                static_for<0, NI / 2>([&](auto pc) {
                    constexpr int pi = decltype(pc)::value, i = 2 * pi;
                    const uint32_t rem = (uint32_t)rem0 + 32u * pi;
                    const uint32_t cb = rem * (uint32_t)npad;                  // (head * 64 + d0) * npad with head * 64 + d0 == rem
                    float x[4], y[4];
                    biased(std::integral_constant<int, i>{}, jc, x);
                    biased(std::integral_constant<int, i + 1>{}, jc, y);
                    if (ok) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            st_h(reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(base) + ((rv + cb + (uint32_t)r * (uint32_t)npad) << 1)), from_f32<T>(x[r]));
                            st_h(reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(base) + ((rv + cb + (uint32_t)(16 + r) * (uint32_t)npad) << 1)), from_f32<T>(y[r]));
                        }
                    }
                });
            });
        }
    }
}

// Tile configuration: BM x BN block tile of BK = 64, WGM x WGN waves (NS_ is unused since the operand rings).
//   Huge  256x256, 4x2 waves (64x128 per wave, 128 accumulator registers), 160 KB LDS, 1 workgroup/CU
//   Mid   128x128, 2x2 waves (64x64 per wave),  80 KB LDS, 2 workgroups/CU
//   Small  64x64,  2x2 waves (32x32 per wave),  40 KB LDS: fills the chip when M*N is small
//   Row    64x512, 1x4 waves (64x128 per wave), 152 KB + 1 KB: fused head1 + row L2-norm
template <int BM_, int BN_, int WGM_, int WGN_, int NS_, int MINW_ = 2>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, NS = NS_, MINW = MINW_;
    static constexpr int NW = WGM * WGN, THREADS = 64 * NW;
    static constexpr int WM = BM / WGM, WN = BN / WGN;          // per-wave sub-tile
    static constexpr int MI = WM / 16, NI = WN / 16;            // 16x16 MFMA tiles per wave
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    static constexpr int A_SPW = BM / 8 / NW, W_SPW = BN / 8 / NW;   // 8-row slabs per wave per stage
    static constexpr int SPW = A_SPW + W_SPW;
    // LDS: separate rings for the two operands -- three K-steps of A, two of W.  The A panel of step k+2
    // stays in flight ACROSS the barrier of step k (counted vmcnt, never 0 in the K-loop): the DMA queue
    // never drains, and only the W half of a step's 2*SPW KB sits on the barrier's critical path.
    static constexpr int A_RING = 3, W_RING = 2;
    static constexpr int LDS = A_RING * A_BYTES + W_RING * W_BYTES;
    // DMA issues of one K-step: Q_B2 ride in the second half of the k-half-1 MFMA block of the previous
    // step, Q_A in the k-half-0 block of the step before they are consumed.  128-accumulator tiles
    // (a K-step is 2000+ cycles) issue everything early; the small-tile configs need the spread, their
    // waves stall ~60 cycles per global_load_lds when two workgroups share the CU's address path.
    static constexpr int Q_A0 = (BM * BN >= 65536) ? 0 : (SPW * NI) / (NI + (NI - NI / 2));
    static constexpr int Q_A = Q_A0 > A_SPW ? A_SPW : Q_A0;      // the W panel (needed first) is always issued early
    static constexpr int Q_B2 = SPW - Q_A;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "slabs must divide evenly over the waves");
};
using CfgMid = TileCfg<128, 128, 2, 2, 2>;
// 160 rows: the pixel x text correlation with the K = 150 ADE20K labels as GEMM rows (10 MFMA row-blocks, 6 % padding instead of
// the 41 % of two 128-row tiles); 92 KB of rings, one workgroup per CU
using CfgLab = TileCfg<160, 128, 2, 2, 2, 1>;
using CfgSmall = TileCfg<64, 64, 2, 2, 2>;
using CfgHuge = TileCfg<256, 256, 4, 2, 2, 1>;     // half the operand bytes per MFMA through the CU's L1 / LDS-DMA path of Mid
// Row config: one workgroup owns complete 512-wide output rows (fused head1 + L2-norm + fp16 casts);
// 128 accumulator registers per lane, one wave per SIMD, 152 KB of rings + 1 KB reduction scratch
using CfgRow = TileCfg<64, 512, 1, 4, 2, 1>;
// (Round 4 measured three more shapes for the ViT block's GEMMs -- 256x128 on 8 waves, 256x128 and 256x256 on 4 waves (one per SIMD, 512
// registers) -- as carriers of a residual prefetch / deferred stores: K-loop alone +11 ... +33 % / +15 ... +22 % / +40 % against this
// 256x256 8-wave tile; removed again, profiles/r04_gemm_experiments.txt.)

// Counted waits go through the BUILTIN, not inline asm: SIInsertWaitcnts understands a pre-existing
// s_waitcnt and keeps its own scoreboard consistent.  An opaque asm wait left it believing that
// epilogue loads were still pending across the persistent-loop back edge, and it planted a
// vmcnt(0) in the middle of every K-step (draining the prefetch: -35% on K=1024 shapes).
// gfx9 simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14]
// Linear tile id -> (row-block, col-block) in GROUPED order: ids run down GROUP_M row-blocks before
// moving to the next column, so the ~64 tiles an XCD works on at any time form a ~8x8 patch
// (A panels + W panels ~ 4 MB = one XCD's L2) instead of 2 full rows of the grid (all of W, which
// does not fit): measured -2.2x L2->fabric fetch on the ViT MLP GEMM (profiles/).
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int& mb, int& nb, int GROUP_M = 8) {
    const int per_group = GROUP_M * tiles_n;
    const int gid = t / per_group;
    const int first_m = gid * GROUP_M;
    const int gsz = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int r = t - gid * per_group;
    mb = first_m + r % gsz;
    nb = r / gsz;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14)); }

// 16 bytes per lane, buffer -> LDS (wave-uniform LDS base + lane * 16), with the buffer's bounds check: offsets >= `bytes` read as zero
__device__ __forceinline__ void buffer_load_lds16(const void* base, int bytes, char* lds, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000),
                                             (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
// two ds_read_b64_tr_b16 = the 8 consecutive k of one MFMA operand lane (k 0-3 from p0, k 4-7 from p1)
__device__ __forceinline__ i32x4_t ds_read_tr16_pair(const char* p0, const char* p1) {
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
    union { v4s_t h[2]; i32x4_t v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)p0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)p1);
    return u.v;
}

// The same two transpose reads as INLINE ASM (lds = the wave-relative LDS byte address, IMM0 / IMM1 immediate offsets).  Through the builtin
// hipcc's s_waitcnt insertion treats every ds_read_b64_tr_b16 as a possible reader of ANY pending direct-to-LDS load and puts
// s_waitcnt vmcnt(0) in front of the first one of each MFMA block: the K-major kernel drained its whole DMA ring twice per K-step (the
// row-major kernels' plain ds_read_b128 do not draw that wait).  The asm form is invisible to that pass -- the K-loop's own counted vmcnt +
// barrier already order the reads behind the loads they need -- and to the lgkmcnt bookkeeping as well: the kernel waits lgkmcnt(0) itself
// before the MFMA block that consumes the fragments (the reads are issued a block earlier, interleaved with that block's first MFMAs).
template <int IMM0, int IMM1>
__device__ __forceinline__ i32x4_t ds_read_tr16_pair_asm(uint32_t lds) {
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    v2i_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(lds), "n"(IMM0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds), "n"(IMM1));
    return i32x4_t{lo[0], lo[1], hi[0], hi[1]};
}

template <typename T, typename CFG, bool CONV, bool RELU_IN, int EPI, int TAG>
__global__ __launch_bounds__(CFG::THREADS, CFG::MINW) void lseg_gemm_kernel(const GemmArgs g) {
    constexpr int BM = CFG::BM, BN = CFG::BN, NS = CFG::NS, NW = CFG::NW;
    constexpr int WM = CFG::WM, WN = CFG::WN, MI = CFG::MI, NI = CFG::NI;
    constexpr int A_BYTES = CFG::A_BYTES, STAGE = CFG::STAGE, A_SPW = CFG::A_SPW, W_SPW = CFG::W_SPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the out-of-line epilogue reads its parameters straight from the kernarg segment
    const GemmArgs& gk = *reinterpret_cast<const GemmArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform -> SALU addressing
    const int wm = w / CFG::WGN, wn = w % CFG::WGN;
    // K-MAJOR operands (TAG == 2; weight gradients dW = dY^T X contract over the token axis, the SLOW axis of both operands as they sit
    // in memory): A is [K, lda] with the M output rows contiguous, W is [K, ldw] with the N output columns contiguous.  An LDS stage
    // holds [64 k][BM] (rows of BM * 2 bytes) instead of [BM][64 k]; it is filled by buffer_load ... lds (16 B / lane, out-of-range rows
    // k >= k_valid come back as ZERO from the buffer's bounds check) and the MFMA fragments are gathered with ds_read_b64_tr_b16
    // (tools/probes/tr_read_probe.py): a 16-lane group reads a [4 k][16 columns] block, lane i receives column i.  Swizzle: the 32-byte
    // segment (16 columns) s of stage row r sits at s ^ gk(r), gk(r) = (r & 3) | ((r >> 3) & 1) << 2 -- the 8 rows a 32-lane read group
    // touches land on 8 different segments = all 64 banks.
    constexpr bool KMAJ = TAG == 2;
    static_assert(!KMAJ || (BM == 128 && BN == 128 && !CONV), "the K-major path is built for 128 x 128 tiles");
    auto gsw = [](int r) { return (r & 3) | (((r >> 3) & 1) << 2); };
    // (the buffer / transpose-read builtins live in __device__ helpers: used directly inside a templated __global__ function they make
    // hipcc drop that instantiation's HOST stub without a diagnostic -- the library then fails to load with an undefined symbol)

    // ---- persistent, XCD-aware tile schedule ------------------------------------------------------
    // Workgroup b lands on XCD b%8 (observed dispatch, used for L2 affinity only).  XCD x owns the
    // contiguous tile range [xs, xs+xc); its gridDim/8 workgroups walk it with stride gridDim/8, so
    // at any time the XCD's workgroups sit on neighbouring tiles (same A row-block, adjacent W
    // panels).  Tiles are n-fastest.
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int per_split = tiles_m * tiles_n;             // output tiles; with split-K every (tile, K-range) pair is a work item
    const int nsplit = g.nsplit > 1 ? g.nsplit : 1;
    const int total = per_split * nsplit;
    const int wpx = gridDim.x >> 3;                       // gridDim.x is a multiple of 8
    int tile, tile_end;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile = xs + idx;
        tile_end = xs + q + (xcd < r ? 1 : 0);
    }
    if (tile >= tile_end) return;

    // ---- load side: two cursors (A runs one K-step ahead of W), both walk (tile, K-step) in order -----
    // per-lane BYTE offsets (32-bit) of this lane's source rows, incl. the swizzled 16-byte chunk
    uint32_t a_off[A_SPW], w_off[W_SPW];
    int w_row[KMAJ ? W_SPW : 1];                          // K-major conv taps: this lane's stage row + the tap's row shift
    const int lrow = lane >> 3;                           // row inside an 8-row slab
    auto setup_a = [&](int t) {
        int mb, nb;
        tile_coords(t % per_split, tiles_m, tiles_n, mb, nb, g.group_m);
        const int m0 = mb * BM;
        if constexpr (KMAJ) {
#pragma unroll
            for (int s = 0; s < A_SPW; ++s) {
                constexpr int LPR = BM / 8, RPI = 64 / LPR;               // lanes (16-byte chunks) per stage row; rows per 1 KB wave instruction
                const int rk = (s * NW + w) * RPI + lane / LPR;           // stage row (k) of this lane
                const int c = (lane % LPR) ^ (gsw(rk) << 1);              // source chunk of LDS chunk slot lane % LPR
                a_off[s] = ((uint32_t)rk * (uint32_t)g.lda + (uint32_t)(m0 + c * 8)) * 2u;
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < A_SPW; ++s) {
            const int r = (s * NW + w) * 8 + lrow;
            int m = m0 + r;
            if (m > g.M - 1) m = g.M - 1;
            uint32_t e;
            if (CONV) {
                const int hw = g.ho * g.wo;
                const int b = m / hw, p = m - b * hw;
                const int y = p / g.wo, x = p - y * g.wo;
                e = (uint32_t)(((b * g.hp + y * g.stride) * g.wp + x * g.stride) * g.cin);
            } else {
                e = (uint32_t)m * (uint32_t)g.lda;
            }
            a_off[s] = (e + (((lane & 7) ^ swz(r)) << 3)) * 2u;
        }
    };
    auto setup_w = [&](int t) {
        int mb, nb;
        tile_coords(t % per_split, tiles_m, tiles_n, mb, nb, g.group_m);
        const int n0 = nb * BN;
        if constexpr (KMAJ) {
            const int tap = g.kconv_cin > 0 ? n0 / g.kconv_cin : 0;
            const int shift = g.kconv_cin > 0 ? (tap / 3 - 1) * g.kconv_wp + (tap % 3 - 1) : 0;
#pragma unroll
            for (int s = 0; s < W_SPW; ++s) {
                constexpr int LPR = BN / 8, RPI = 64 / LPR;
                const int rk = (s * NW + w) * RPI + lane / LPR;
                const int c = (lane % LPR) ^ (gsw(rk) << 1);
                if (g.kconv_cin > 0) {      // conv taps: the row is resolved per load (it may fall outside the map: zero)
                    w_row[s] = rk + shift;
                    w_off[s] = (uint32_t)(n0 - tap * g.kconv_cin + c * 8) * 2u;
                } else {
                    w_off[s] = ((uint32_t)rk * (uint32_t)g.ldw + (uint32_t)(n0 + c * 8)) * 2u;
                }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < W_SPW; ++s) {
            const int r = (s * NW + w) * 8 + lrow;
            int n = n0 + r;
            if (n > g.N - 1) n = g.N - 1;
            w_off[s] = ((uint32_t)n * (uint32_t)g.ldw + (((lane & 7) ^ swz(r)) << 3)) * 2u;
        }
    };

    const int nk = g.K >> 6;
    const int nseg = g.split ? 3 : 1;           // split precision: K-loop segments A_hi.W_hi, A_lo.W_hi, A_hi.W_lo
    const int cpt = CONV ? (g.cin >> 6) : 1;    // 64-wide K chunks per conv tap
    constexpr int W_RING_OFF = CFG::A_RING * A_BYTES;
    // After the last tile a cursor keeps re-issuing that tile (never consumed): every step issues exactly
    // SPW loads per wave, so the counted vmcnt stays exact.
    // K-step range of a work item: everything without split-K, [s * split_steps, ...) with it
    auto k_beg = [&](int t) { return nsplit > 1 ? (t / per_split) * g.split_steps : 0; };
    auto k_end = [&](int t) { const int e = nsplit > 1 ? (t / per_split + 1) * g.split_steps : nk; return e < nk ? e : nk; };
    int atile = tile, akt = k_beg(tile), aend = k_end(tile), aslot = 0, aseg = 0;
    int wtile = tile, wkt = akt, wend = aend, wslot = 0, wseg = 0;
    setup_a(atile);
    setup_w(wtile);
    const char *abase = nullptr, *wbase = nullptr;        // source bases of the K-steps the cursors point at
    auto a_src = [&]() {
        int koff_a;
        if (CONV) {
            const int tap = akt / cpt, ci0 = (akt - tap * cpt) << 6;
            const int ky = tap / 3, kx = tap - ky * 3;
            koff_a = (ky * g.wp + kx) * g.cin + ci0;
        } else {
            koff_a = akt << 6;
        }
        return reinterpret_cast<const char*>(g.A + (aseg == 1 ? g.a_plane : 0) + koff_a);
    };
    auto advance_a = [&]() {
        aslot = aslot + 1 == CFG::A_RING ? 0 : aslot + 1;
        if (++akt == aend) {
            if (++aseg == nseg) {
                aseg = 0;
                if (atile + wpx < tile_end) atile += wpx;
                setup_a(atile);
                aend = k_end(atile);
            }
            akt = k_beg(atile);
        }
    };
    auto advance_w = [&]() {
        wslot ^= 1;
        if (++wkt == wend) {
            if (++wseg == nseg) {
                wseg = 0;
                if (wtile + wpx < tile_end) wtile += wpx;
                setup_w(wtile);
                wend = k_end(wtile);
            }
            wkt = k_beg(wtile);
        }
    };
    // One issue GROUP = the W panel at the W cursor, then the A panel at the A cursor (one K-step further on);
    // q indexes the group's SPW direct-to-LDS loads in that order.  A group may be split over two blocks of
    // MFMAs: the bases are latched on the first load of each kind, the cursors advance on the last.
    auto issue_q = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < W_SPW) {
            if constexpr (KMAJ) {
                if (g.kconv_cin > 0) {
                    const int r = w_row[q] + (wkt << 6);
                    const uint32_t vo = (r < 0 || r >= g.k_valid) ? 0xffffffffu : (uint32_t)r * (uint32_t)g.ldw * 2u + w_off[q];
                    buffer_load_lds16(g.W, (int)((size_t)g.k_valid * g.ldw * 2), smem + W_RING_OFF + wslot * CFG::W_BYTES + (q * NW + w) * 1024, vo, 0u);
                } else {
                    buffer_load_lds16(g.W, (int)((size_t)g.k_valid * g.ldw * 2), smem + W_RING_OFF + wslot * CFG::W_BYTES + (q * NW + w) * 1024,
                                      w_off[q], (uint32_t)wkt * 128u * (uint32_t)g.ldw);
                }
            } else {
                if constexpr (q == 0) wbase = reinterpret_cast<const char*>(g.W + (wseg == 2 ? g.w_plane : 0) + (wkt << 6));
                glds_slab_off(wbase, w_off[q], smem + W_RING_OFF + wslot * CFG::W_BYTES + (q * NW + w) * 1024);
            }
            if constexpr (q == W_SPW - 1) advance_w();
        } else {
            constexpr int qa = q - W_SPW;
            if constexpr (KMAJ) {
                buffer_load_lds16(g.A, (int)((size_t)g.k_valid * g.lda * 2), smem + aslot * A_BYTES + (qa * NW + w) * 1024,
                                  a_off[qa], (uint32_t)akt * 128u * (uint32_t)g.lda);
            } else {
                if constexpr (qa == 0) abase = a_src();
                glds_slab_off(abase, a_off[qa], smem + aslot * A_BYTES + (qa * NW + w) * 1024);
            }
            if constexpr (qa == A_SPW - 1) advance_a();
        }
    };

    // ---- MFMA side: software-pipelined K-loop ---------------------------------------------------------
    // One workgroup barrier per K-step, in the MIDDLE of the k-half-1 MFMA block:
    //   block A   k-half-0 MFMAs of step k        | reads the k-half-1 fragments of stage k%2
    //   block B1  first half of the k-half-1 MFMAs
    //   vmcnt(0) ; s_barrier       -> every wave's DMA of step k+1 has landed, every wave is done reading stage k%2
    //   block B2  rest of the k-half-1 MFMAs      | reads the k-half-0 fragments of step k+1 (other stage)
    //                                             | issues the DMA of step k+2 into stage k%2
    // so the wave always leaves the barrier with 8+ MFMAs of fragments in registers (no post-barrier
    // LDS ramp: that ramp plus barrier skew idled the matrix pipe ~25% of every K-step in the
    // barrier-at-the-top loop, tools/gemm_phase_probe.py), and a DMA has a full K-step to land.
    f32x4_t acc[NI][MI];
    i32x4_t wf0[NI], af0[MI], wf1[NI], af1[MI];
    // LDS fragment byte offsets of this lane (ks = 0 / 1); sub-tiles add i*2048 / j*2048
    const int frow = lane & 15;
    const int foff0 = tile_off(frow, lane >> 4), foff1 = tile_off(frow, 4 + (lane >> 4));
    const int wbase_off = wn * WN * 128, abase_off = wm * WM * 128;   // inside a W / A ring slot
#ifdef GEMM_ABL_NOREAD      // ablation builds (tools/): K-loop without LDS fragment reads / DMA issues / MFMAs
    auto ldfrag = [&](const char* p) { return i32x4_t{lane, lane, lane, lane}; };
#else
    auto ldfrag = [&](const char* p) { return *reinterpret_cast<const i32x4_t*>(p); };
#endif
#ifdef GEMM_ABL_NOMFMA
    auto mm = [&](i32x4_t a, i32x4_t b, f32x4_t c) { asm volatile("" ::"v"(a), "v"(b)); return c; };
#else
    auto mm = [&](i32x4_t a, i32x4_t b, f32x4_t c) { return mfma16<T>(a, b, c); };
#endif
    // K-major stages: per-lane offsets of the [4 k][16 col] transpose-read blocks of each 16-wide sub-tile (loop invariant); khalf and
    // the two 4-row halves of an 8-k operand are immediates
    int a_tr[KMAJ ? MI : 1], w_tr[KMAJ ? NI : 1];
    if constexpr (KMAJ) {
        const int ti = lane & 15, kg = lane >> 4, gl = (ti >> 2) | ((kg & 1) << 2);
#pragma unroll
        for (int j = 0; j < MI; ++j) a_tr[j] = (kg * 8 + (ti >> 2)) * (BM * 2) + ((((wm * WM) >> 4) + j) ^ gl) * 32 + (ti & 3) * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) w_tr[i] = (kg * 8 + (ti >> 2)) * (BN * 2) + ((((wn * WN) >> 4) + i) ^ gl) * 32 + (ti & 3) * 8;
    }
    // K-major fragments through the asm transpose reads (see ds_read_tr16_pair_asm) unless a ReLU runs on the fragments right after the read
    // (that form needs the compiler's own lgkmcnt tracking; the training step hands in materialised ReLU maps and never takes it)
    constexpr bool TR_ASM = KMAJ && !RELU_IN;
    const uint32_t lds0 = TR_ASM ? (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem) : 0u;
    auto ldfrag_t = [&](const char* st, int off, auto khc, auto pitchc) {
        constexpr int khalf = decltype(khc)::value, pitch = decltype(pitchc)::value;
        if constexpr (TR_ASM) return ds_read_tr16_pair_asm<(khalf * 32) * pitch, (khalf * 32 + 4) * pitch>(lds0 + (uint32_t)(st - smem) + (uint32_t)off);
        else return ds_read_tr16_pair(st + off + (khalf * 32) * pitch, st + off + (khalf * 32 + 4) * pitch);
    };
    // khc = K-half of the 64-deep stage (std::integral_constant 0 / 1)
    auto lda = [&](const char* st, int j, auto khc) {
        constexpr int kh = decltype(khc)::value;
        if constexpr (KMAJ) return ldfrag_t(st, a_tr[j], khc, std::integral_constant<int, BM * 2>{});
        i32x4_t v = ldfrag(st + abase_off + j * 2048 + (kh ? foff1 : foff0));
        if (RELU_IN) v = relu_frag(v);
        return v;
    };
    auto ldw = [&](const char* st, int i, auto khc) {
        constexpr int kh = decltype(khc)::value;
        if constexpr (KMAJ) {
            i32x4_t v = ldfrag_t(st, w_tr[i], khc, std::integral_constant<int, BN * 2>{});
            if (RELU_IN) v = relu_frag(v);                    // K-major conv wgrad: the W operand is the conv input, read through a ReLU
            return v;
        }
        return ldfrag(st + wbase_off + i * 2048 + (kh ? foff1 : foff0));
    };
    using kh0_t = std::integral_constant<int, 0>;
    using kh1_t = std::integral_constant<int, 1>;

    unsigned long long tmark[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // dbg&4: cycles in {vm wait, barrier, epilogue, MFMA blocks}
    unsigned long long kc0 = 0, kr0 = 0;                       // dbg&4: whole-kernel s_memtime / s_memrealtime (100 MHz)
    if ((g.dbg & 4)) { kc0 = __builtin_readcyclecounter(); kr0 = __builtin_amdgcn_s_memrealtime(); }

    int ca = 0, cw = 0;                                  // ring slots of the K-step being computed
    const bool early = NW == 8 && w >= 4 && !(g.dbg & 8);   // see sync() below (dbg bit 3: A/B switch, tools)
    auto step = [&]() {
        const char* sa = smem + ca * A_BYTES;                                   // this step's panels
        const char* sw = smem + W_RING_OFF + cw * CFG::W_BYTES;
        const int na = ca + 1 == CFG::A_RING ? 0 : ca + 1;
        const char* nxa = smem + na * A_BYTES;                                   // next step's
        const char* nxw = smem + W_RING_OFF + (cw ^ 1) * CFG::W_BYTES;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if ((g.dbg & 4)) t0 = __builtin_readcyclecounter();
        // Memory operations are interleaved ONE per MFMA (fenced): a wave that runs alone on its SIMD -- its
        // partner waiting at the barrier -- then keeps the matrix pipe fed; in groups of 4 MFMAs followed by
        // 3-5 loads the pipe drained after every group (a lone wave reached ~65% of the MFMA rate).
        // ---- block A: NI*MI MFMAs; ops = k-half-1 fragment reads (A fragments first), then Q_A DMA issues
        if constexpr (TR_ASM) {                        // the k-half-0 fragments (asm reads of the previous block B2 / the prologue) have arrived
            wait_lgkmcnt0();
            __builtin_amdgcn_sched_barrier(0);         // (no MFMA of this block may be scheduled above the wait: the compiler sees no dependence)
        }
        {
            constexpr int NMF = NI * MI, OPS = NI + MI + CFG::Q_A;
            static_for<0, NMF>([&](auto tc) {
                constexpr int t = decltype(tc)::value, i = t / MI, j = t % MI;
                acc[i][j] = mm(wf0[i], af0[j], acc[i][j]);
                static_for<(t * OPS) / NMF, ((t + 1) * OPS) / NMF>([&](auto oc) {
                    constexpr int o = decltype(oc)::value;
                    if constexpr (o < MI) af1[o] = lda(sa, o, kh1_t{});
                    else if constexpr (o < MI + NI) wf1[o - MI] = ldw(sw, o - MI, kh1_t{});
#ifndef GEMM_ABL_NODMA
                    else issue_q(std::integral_constant<int, CFG::Q_B2 + o - MI - NI>{});
#endif
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // The step's barrier: every wave's reads of this step's ring slots are complete (all issued in block A)
        // and its share of step k+1 has landed; only the A panel of step k+2 (the last A_SPW loads issued) stays
        // in flight across it.  8-wave workgroups put two waves on every SIMD (w and w+4): the second group
        // takes the barrier one block EARLIER in its instruction stream, which shifts it a quarter step in time --
        // while one wave of a SIMD is in the load-heavy block B2 its partner is in the MFMA-only block B1.
        auto sync = [&]() {
            if ((g.dbg & 4)) t1 = __builtin_readcyclecounter();
            wait_lgkmcnt0();
            wait_vmcnt<A_SPW>();
            if ((g.dbg & 4)) t2 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();
            if ((g.dbg & 4)) t3 = __builtin_readcyclecounter();
        };
        if (NW == 8 && early) sync();
        if constexpr (TR_ASM) {                        // the k-half-1 fragments (asm reads of block A) have arrived
            wait_lgkmcnt0();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- block B1
        static_for<0, NI / 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = mm(wf1[i], af1[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!(NW == 8 && early)) sync();
        // ---- block B2
        {   // (NI - NI/2)*MI MFMAs; ops = next step's k-half-0 fragment reads (A first), then Q_B2 DMA issues
            constexpr int I0 = NI / 2, NMF = (NI - I0) * MI, OPS = NI + MI + CFG::Q_B2;
            static_for<0, NMF>([&](auto tc) {
                constexpr int t = decltype(tc)::value, i = I0 + t / MI, j = t % MI;
                acc[i][j] = mm(wf1[i], af1[j], acc[i][j]);
                static_for<(t * OPS) / NMF, ((t + 1) * OPS) / NMF>([&](auto oc) {
                    constexpr int o = decltype(oc)::value;
                    if constexpr (o < MI) af0[o] = lda(nxa, o, kh0_t{});
                    else if constexpr (o < MI + NI) wf0[o - MI] = ldw(nxw, o - MI, kh0_t{});
#ifndef GEMM_ABL_NODMA
                    else issue_q(std::integral_constant<int, o - MI - NI>{});
#endif
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        ca = na; cw ^= 1;
        if ((g.dbg & 4)) {
            const unsigned long long t4 = __builtin_readcyclecounter();
            tmark[0] += t2 - t1; tmark[1] += t3 - t2; tmark[3] += (t1 - t0) + (t4 - t3);
        }
    };

    // prologue.  Steady state at the barrier of step k: landed W(k+1), A(k+1); in flight A(k+2); the group
    // {W(k+2), A(k+3)} is issued behind the barrier.  So: W(0), A(0) | A(1) | {W(1), A(2)} (its tail in block A).
    static_for<0, W_SPW>([&](auto qc) { issue_q(qc); });                                   // W(0)
    static_for<0, A_SPW>([&](auto qc) { issue_q(std::integral_constant<int, W_SPW + decltype(qc)::value>{}); });   // A(0)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < NI; ++i) wf0[i] = ldw(smem + W_RING_OFF, i, kh0_t{});
#pragma unroll
    for (int j = 0; j < MI; ++j) af0[j] = lda(smem, j, kh0_t{});
    static_for<0, A_SPW>([&](auto qc) { issue_q(std::integral_constant<int, W_SPW + decltype(qc)::value>{}); });   // A(1)
    static_for<0, CFG::Q_B2>([&](auto qc) { issue_q(qc); });                               // head of {W(1), A(2)}

    constexpr bool BIAS_PREFETCH = MI * NI <= 16;
    float4 biasv[EPI != EPI_GENERIC ? NI : 1];
    bool res_loaded = false;                   // GEMM_RES_PRELOAD: acc already holds the next tile's residual
    for (; tile < tile_end; tile += wpx) {
        int pm0, pn0;
        {   // this tile's origin and, for the specialised epilogues, its bias (consumed after the K-loop)
            int mbc, nbc;
            tile_coords(tile % per_split, tiles_m, tiles_n, mbc, nbc, g.group_m);
            pm0 = mbc * BM; pn0 = nbc * BN;
            if constexpr (EPI != EPI_GENERIC && EPI != EPI_PART32 && BIAS_PREFETCH) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    biasv[i] = *reinterpret_cast<const float4*>(g.bias + (EPI == EPI_PIX16 ? (pn0 + wn * WN + i * 16) % g.ps_C : pn0 + wn * WN + i * 16) + (lane >> 4) * 4);
            }
        }
        constexpr bool RES_PRE = GEMM_RES_PRELOAD && EPI == EPI_RES32 && !CONV && CFG::BN != 512;
        if constexpr (RES_PRE) {
            // the accumulators start from the residual: loaded by the previous tile's epilogue, or here for the workgroup's first tile
            if (!res_loaded) res_preload<MI, NI>(g, acc, pm0 + wm * WM, pn0 + wn * WN, lane);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        for (int kt = (k_end(tile) - k_beg(tile)) * nseg; kt > 0; --kt) step();
        // ---- epilogue: the next tile's first K-step is already in flight / in registers
        unsigned long long te0 = 0;
        if ((g.dbg & 4)) te0 = __builtin_readcyclecounter();
        {
            const int m0c = pm0, n0c = pn0;

        if constexpr (CFG::BN == 512) {
            // ---- fused head (lseg_net.py:185-194): v = head1(x)+bias ; a = fp16(s * fp16(v/||v||)) ----
            // lane owns, for each of its MI rows, NI groups of 4 columns inside this wave's 128
            // columns; the row's sum of squares is reduced over the 4 lane groups (shuffles) and
            // over the 4 waves (LDS scratch behind the stage ring), in a fixed order.
            float* red = reinterpret_cast<float*>(smem + CFG::LDS);      // [4 waves][64 rows]
            const int n_first = n0c + wn * WN + (lane >> 4) * 4;
            float ssq[MI];
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                float s_ = 0.f;
                static_for<0, NI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const float4 bv = *reinterpret_cast<const float4*>(g.bias + n_first + i * 16);
                    acc[i][j][0] += bv.x; acc[i][j][1] += bv.y; acc[i][j][2] += bv.z; acc[i][j][3] += bv.w;
                    s_ += acc[i][j][0] * acc[i][j][0] + acc[i][j][1] * acc[i][j][1] +
                          acc[i][j][2] * acc[i][j][2] + acc[i][j][3] * acc[i][j][3];
                });
                s_ += __shfl_xor(s_, 16);
                s_ += __shfl_xor(s_, 32);
                ssq[j] = s_;
            });
            if (lane < 16) {
#pragma unroll
                for (int j = 0; j < MI; ++j) red[wn * 64 + j * 16 + lane] = ssq[j];
            }
            __syncthreads();
            static_for<0, MI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int rl = j * 16 + (lane & 15);
                const float nrm = sqrtf((red[rl] + red[64 + rl]) + (red[128 + rl] + red[192 + rl]));
                const int m = m0c + rl;
                const int r16 = lane >> 4;
                static_for<0, NI / 2>([&](auto pc) {
                    constexpr int i = 2 * decltype(pc)::value;
                    float y[2][4];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[q][r] = g.rn_scale * round_f16(acc[i + q][j][r] / nrm);
                    const uint32_t a0 = pack2<F16>(y[0][0], y[0][1]), a1 = pack2<F16>(y[0][2], y[0][3]);
                    const uint32_t b0 = pack2<F16>(y[1][0], y[1][1]), b1 = pack2<F16>(y[1][2], y[1][3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    const int col = n0c + wn * WN + i * 16 + (r16 & 1) * 16 + (r16 >> 1) * 8;
                    if (m < g.M)
                        *reinterpret_cast<uint4*>((uint16_t*)g.C + (size_t)m * g.ldc + col) =
                            make_uint4(s0[0], s1[0], s0[1], s1[1]);
                });
            });
        } else if constexpr (EPI != EPI_GENERIC) {
            if constexpr (!BIAS_PREFETCH && EPI != EPI_PART32) {       // 128-accumulator tiles have no registers to spare across the K-loop
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    biasv[i] = *reinterpret_cast<const float4*>(g.bias + (EPI == EPI_PIX16 ? (n0c + wn * WN + i * 16) % g.ps_C : n0c + wn * WN + i * 16) + (lane >> 4) * 4);
                // ONE explicit wait (a builtin: the compiler's scoreboard sees it).  Left to the compiler, an epilogue whose stores sit in
                // per-row conditional blocks (EPI_QKV16) got an s_waitcnt vmcnt(0) at the top of EVERY block -- "the bias may still be
                // pending on the path that skipped the previous block" -- and each of those also waits for the previous block's stores:
                // four store round trips per tile, attn.qkv 197 -> 223 us in the engine at B = 36 (profiles/r04_gemm_experiments.txt).
                wait_vmcnt<0>();
            }
            if constexpr (GEMM_EPI_ABL == 2) {          // attribution build: the K-loop alone (accumulators kept live)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
            } else if constexpr (RES_PRE) {
                const bool next = tile + wpx < tile_end;
                int nmb = 0, nnb = 0;
                if (next) tile_coords((tile + wpx) % per_split, tiles_m, tiles_n, nmb, nnb, g.group_m);
                res_store_and_preload<MI, NI>(g, acc, m0c + wm * WM, n0c + wn * WN, lane, biasv, next, nmb * BM + wm * WM, nnb * BN + wn * WN);
                res_loaded = next;
            } else {
                fast_epilogue<T, EPI, MI, NI>(g, gk, acc, m0c + wm * WM, n0c + wn * WN, lane, biasv,
                                              nsplit > 1 ? (size_t)(tile / per_split) * g.c_split_stride : 0);
            }
        } else {
        int ncol[NI];
        ColPart cp[NI], cpw[NI / 2];
        float4 bias[NI];
        const bool wide = epilogue_cols<NI>(g, n0c + wn * WN + (lane >> 4) * 4, lane, ncol, cp, cpw, bias);
        // runtime loop over the MI output rows of this lane (one copy of the epilogue code); the
        // accumulator row is selected with static indices so acc[][] stays in registers
#pragma unroll 1
        for (int j = 0; j < MI; ++j) {
            f32x4_t row[NI];
            static_for<0, MI>([&](auto jc) {
                constexpr int js = decltype(jc)::value;
                if (j == js) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) row[i] = acc[i][js];
                }
            });
            const int m = m0c + wm * WM + j * 16 + (lane & 15);
            if ((g.dbg & 3) == 2) { asm volatile("" ::"v"(row[0][0]), "v"(row[NI - 1][3])); continue; }
            if (m < g.M) epilogue_row<T, NI>(g, m, ncol, cp, cpw, wide, bias, row, nsplit > 1 ? (size_t)(tile / per_split) * g.c_split_stride : 0);
        }
        }
        }
        if ((g.dbg & 4)) tmark[2] += __builtin_readcyclecounter() - te0;
    }
    if ((g.dbg & 4) && lane == 0 && g.res2) {
        unsigned long long* o = (unsigned long long*)g.res2 + ((size_t)blockIdx.x * NW + w) * 8;
        o[0] = tmark[0]; o[1] = tmark[1]; o[2] = tmark[2]; o[3] = tmark[3]; o[4] = 0; o[5] = 0;
        o[6] = __builtin_readcyclecounter() - kc0;              // whole kernel, shader cycles
        o[7] = __builtin_amdgcn_s_memrealtime() - kr0;          // whole kernel, 100 MHz ticks
    }
    wait_vmcnt<0>();     // drain the never-consumed tail loads before the LDS is released
}

template <typename T, typename CFG, bool CONV, bool RELU_IN, int EPI, int TAG>
int launch_one(const GemmArgs& g, hipStream_t stream) {
    const int tiles = ((g.M + CFG::BM - 1) / CFG::BM) * ((g.N + CFG::BN - 1) / CFG::BN) * (g.nsplit > 1 ? g.nsplit : 1);
    const size_t lds = CFG::LDS + (CFG::BN == 512 ? 1024 : 0);
    // persistent grid: a multiple of 8 (one slice per XCD), at most `slots` resident workgroups
    constexpr int by_lds = 163840 / CFG::LDS, by_waves = 8 / (CFG::NW / 4);
    constexpr int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
    int dev = 0;
    LSEG_HIP_TRY(hipGetDevice(&dev));
    const int slots = device_cu_count(dev) * per_cu;
    int grid = ((tiles + 7) / 8) * 8;
    if (grid > slots) grid = slots;
    if (g.max_grid >= 8 && grid > g.max_grid) grid = g.max_grid & ~7;      // tools: part of the chip (tools/epilogue_table.py)
    auto kern = lseg_gemm_kernel<T, CFG, CONV, RELU_IN, EPI, TAG>;
    // the dynamic-LDS opt-in is a per-DEVICE function attribute: one bit per device (a process may drive several GPUs, one
    // engine and one host thread each: additional_utils/models.py:229-238), set with an atomic so concurrent threads are safe
    static std::atomic<unsigned long long> attr_done{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds, stream, g);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T, bool CONV, bool RELU_IN, int EPI, int TAG>
int pick_tile(const GemmArgs& g, hipStream_t stream) {
    const long t_mid = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.nsplit > 1 ? g.nsplit : 1);
    // tools/tests: 1 = 64x64, 2 = 128x128, 6 = 256x256.  Read ONCE per process (function-local static: thread-safe initialisation, no getenv
    // on the launch path -- several hundred launches per forward, possibly from several host threads); the sweeping tool and the forced-tile
    // tests start one interpreter per tile (tools/tile_sweep.py, tests/test_gpu_tile_configs.py)
    static const int force = [] { const char* e = getenv("LSEG_GEMM_TILE"); return e ? atoi(e) : 0; }();
    int pick = t_mid >= 192 ? 2 : 1;
    if (EPI == EPI_GENERIC && !CONV && g.map_mode == MAP_LABELPLANES && g.M > 128 && g.M <= 160 && !force)
        return launch_one<T, CfgLab, false, false, EPI_GENERIC, TAG>(g, stream);
    if (EPI != EPI_GENERIC && pick == 2) {
        // 256x256 tiles (one 8-wave workgroup per CU) move half the operand bytes per MFMA through the
        // CU's L1/LDS-DMA path and run ~12% faster per flop than two 128x128 workgroups -- when the
        // tile count quantises well over the 256 CUs.  Persistent schedules: ceil(tiles / slots) rounds.
        const long t_huge = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * (g.nsplit > 1 ? g.nsplit : 1);
        const double time_huge = (double)((t_huge + 255) / 256) * 4.0 / 1.12;
        const double time_mid = (double)((t_mid + 511) / 512) * 2.0;
        if (time_huge < time_mid) pick = 6;
    }
    if (g.tile_hint == 2 || g.tile_hint == 6) pick = g.tile_hint;
    if (force) pick = force;
    if (pick == 6 && (g.N % 256) != 0) pick = 2;      // the specialised epilogues write whole tile rows: N must be a multiple of BN
    if constexpr (EPI != EPI_GENERIC) {
        if (pick == 6) return launch_one<T, CfgHuge, CONV, RELU_IN, EPI, TAG>(g, stream);
    }
    if (pick == 6) pick = 2;
    if (pick == 2) return launch_one<T, CfgMid, CONV, RELU_IN, EPI, TAG>(g, stream);
    return launch_one<T, CfgSmall, CONV, RELU_IN, EPI, TAG>(g, stream);
}

// Which specialised epilogue (if any) reproduces this launch exactly.
template <typename T>
int select_epi(const GemmArgs& g) {
    constexpr int dt = std::is_same<T, BF16>::value ? DT_BF16 : DT_F16;
    static const bool off = getenv("LSEG_GEMM_GENERIC_EPI") != nullptr;       // A/B switch (tools)
    if (!off && (g.nsplit > 1 || g.kmajor) && !g.split && !(g.dbg & 3) && !g.bias && !g.round_mid && (g.N % 128) == 0 && g.map_mode == MAP_LINEAR &&
        g.res_mode == RES_NONE && g.act == ACT_NONE && g.out_dtype == DT_F32 && (g.ldc % 4) == 0 && (g.c_split_stride % 4) == 0 &&
        !(reinterpret_cast<uintptr_t>(g.C) & 15) && !(g.conv && g.relu_in))
        return EPI_PART32;
    if (off || g.split || g.nsplit > 1 || (g.dbg & 3) || !g.bias || g.round_mid || (g.N % 128) != 0) return EPI_GENERIC;
    if (g.map_mode == MAP_PIXSHUF && !g.conv && g.bias_mod == g.ps_C && (g.ps_C % 32) == 0 && g.N == g.ps_s * g.ps_s * g.ps_C && g.out_dtype == dt &&
        g.res_mode == RES_NONE && g.act == ACT_NONE && !(reinterpret_cast<uintptr_t>(g.C) & 15) && !(reinterpret_cast<uintptr_t>(g.bias) & 15))
        return EPI_PIX16;
    if (g.bias_mod) return EPI_GENERIC;
    if (g.res2 && !(g.dbg & 4) && g.map_mode != MAP_PADDED) return EPI_GENERIC;
    if ((reinterpret_cast<uintptr_t>(g.C) & 15) || (reinterpret_cast<uintptr_t>(g.bias) & 15)) return EPI_GENERIC;
    if (g.map_mode == MAP_PADDED && g.out_dtype == dt && (g.ldc % 8) == 0 && (g.act == ACT_NONE || g.act == ACT_RELU) &&
        (g.res_mode == RES_NONE || (g.res_mode == RES_DEST && g.res_dtype == dt &&
                                    !((reinterpret_cast<uintptr_t>(g.res) | reinterpret_cast<uintptr_t>(g.res2)) & 7))))
        return EPI_PAD16;
    if (g.conv) return EPI_GENERIC;
    if (g.map_mode == MAP_LINEAR && g.res_mode == RES_NONE && g.out_dtype == dt && (g.ldc % 8) == 0) {
        const bool al = !((reinterpret_cast<uintptr_t>(g.C_pre) | reinterpret_cast<uintptr_t>(g.dgelu_pre)) & 15);
        if (g.act == ACT_NONE && g.dgelu_pre && !g.C_pre && al) return EPI_LIN16_DGELU;
        if (g.act == ACT_GELU && g.C_pre && !g.dgelu_pre && al) return EPI_LIN16_GELU2;
        if (g.C_pre || g.dgelu_pre) return EPI_GENERIC;
        if (g.act == ACT_NONE) return EPI_LIN16;
        if (g.act == ACT_GELU) return EPI_LIN16_GELU;
        return EPI_GENERIC;
    }
    if (g.map_mode == MAP_LINEAR && g.res_mode == RES_NONE && dt == DT_BF16 && g.out_dtype == DT_F16 && (g.ldc % 8) == 0 && g.act == ACT_NONE)
        return EPI_LIN16_F16;
    if (g.map_mode == MAP_LINEAR && g.res_mode == RES_NONE && g.out_dtype == DT_F32 && g.act == ACT_NONE && (g.ldc % 4) == 0 && !g.C_pre && !g.dgelu_pre)
        return EPI_LIN32;
    if (g.map_mode == MAP_LINEAR && g.res_mode == RES_DEST && g.out_dtype == DT_F32 && g.res_dtype == DT_F32 &&
        g.act == ACT_NONE && (g.ldc % 4) == 0 && !(reinterpret_cast<uintptr_t>(g.res) & 15))
        return EPI_RES32;
    if (g.map_mode == MAP_QKV && g.res_mode == RES_NONE && g.act == ACT_NONE && g.out_dtype == dt &&
        (g.qkv_dim % 128) == 0 && g.qkv_dim == g.qkv_heads * 64 && g.qkv_ntok > 0 &&       // a wave's (<= 128) columns inside one of q | k | v
        (long long)((g.M + g.qkv_ntok - 1) / g.qkv_ntok) * g.qkv_heads * g.qkv_npad * 64 < (1ll << 31) &&      // 32-bit element offsets in the epilogue
        !((reinterpret_cast<uintptr_t>(g.Ck) | reinterpret_cast<uintptr_t>(g.Cv)) & 15))
        return EPI_QKV16;
    return EPI_GENERIC;
}

template <typename T>
int dispatch(const GemmArgs& g, hipStream_t stream) {
    if (g.kmajor) {
        if (select_epi<T>(g) != EPI_PART32 || g.conv || (g.relu_in && !g.kconv_cin) || (g.kconv_cin && ((g.kconv_cin % 128) || g.ldw != g.kconv_cin)) || g.k_valid < 1 || g.k_valid > g.K || g.K - g.k_valid >= 64 ||
            (double)g.k_valid * g.lda * 2 >= 2.0e9 || (double)g.k_valid * g.ldw * 2 >= 2.0e9)
            return set_error(LSEG_ERR_UNSUPPORTED, "K-major GEMM: needs fp32 slab output (MAP_LINEAR, no bias / residual), N %% 128 == 0, operands < 2 GB");
        if (g.relu_in) return launch_one<T, CfgMid, false, true, EPI_PART32, 2>(g, stream);
        return launch_one<T, CfgMid, false, false, EPI_PART32, 2>(g, stream);
    }
    if (g.map_mode == MAP_ROWNORM) {
        if (g.split) return set_error(LSEG_ERR_UNSUPPORTED, "the fused head has no split-precision form (use the two-kernel path)");
        if (g.conv || g.relu_in || g.N != 512 || !g.bias)
            return set_error(LSEG_ERR_UNSUPPORTED, "fused head needs a plain GEMM with N == 512 and a bias");
        return launch_one<T, CfgRow, false, false, EPI_GENERIC, 0>(g, stream);
    }
    const int epi = select_epi<T>(g);
    if (g.conv) {
        if (epi == EPI_PART32) return pick_tile<T, true, false, EPI_PART32, 0>(g, stream);
        if (epi == EPI_PAD16) {
            if (g.relu_in) return pick_tile<T, true, true, EPI_PAD16, 0>(g, stream);
            return pick_tile<T, true, false, EPI_PAD16, 0>(g, stream);
        }
        if (g.relu_in) return pick_tile<T, true, true, EPI_GENERIC, 0>(g, stream);
        return pick_tile<T, true, false, EPI_GENERIC, 0>(g, stream);
    }
    if (g.relu_in && !g.kmajor) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: relu_in is only implemented for the conv path");
    if (g.split && !std::is_same<T, F16>::value) return set_error(LSEG_ERR_UNSUPPORTED, "split precision runs on fp16 (hi, lo) pairs");
    switch (epi) {
        case EPI_PAD16: return pick_tile<T, false, false, EPI_PAD16, 0>(g, stream);
        case EPI_LIN16: return pick_tile<T, false, false, EPI_LIN16, 0>(g, stream);
        case EPI_LIN16_GELU:
            if (g.tag == 1) return pick_tile<T, false, false, EPI_LIN16_GELU, 1>(g, stream);
            return pick_tile<T, false, false, EPI_LIN16_GELU, 0>(g, stream);
        case EPI_RES32: return pick_tile<T, false, false, EPI_RES32, 0>(g, stream);
        case EPI_LIN16_F16: return pick_tile<T, false, false, EPI_LIN16_F16, 0>(g, stream);
        case EPI_LIN32: return pick_tile<T, false, false, EPI_LIN32, 0>(g, stream);
        case EPI_LIN16_GELU2: return pick_tile<T, false, false, EPI_LIN16_GELU2, 0>(g, stream);
        case EPI_LIN16_DGELU: return pick_tile<T, false, false, EPI_LIN16_DGELU, 0>(g, stream);
        case EPI_PIX16: return pick_tile<T, false, false, EPI_PIX16, 0>(g, stream);
        case EPI_QKV16: return pick_tile<T, false, false, EPI_QKV16, 0>(g, stream);
        case EPI_PART32: return pick_tile<T, false, false, EPI_PART32, 0>(g, stream);
        default: return pick_tile<T, false, false, EPI_GENERIC, 0>(g, stream);
    }
}

}  // namespace

bool gemm_epilogue_is_pad16(const GemmArgs& g, int ab_dtype) {
    if (ab_dtype == DT_BF16) return select_epi<BF16>(g) == EPI_PAD16;
    if (ab_dtype == DT_F16) return select_epi<F16>(g) == EPI_PAD16;
    return false;
}

bool gemm_qkv_scales_q(const GemmArgs& g, int ab_dtype) {
    const int e = ab_dtype == DT_BF16 ? select_epi<BF16>(g) : ab_dtype == DT_F16 ? select_epi<F16>(g) : EPI_GENERIC;
    return !g.conv && !g.kmajor && g.map_mode == MAP_QKV && e == EPI_QKV16;
}

bool gemm_fuses_gelu(const GemmArgs& g, int ab_dtype) {
    static const bool off = getenv("LSEG_NO_GELU_FUSE") != nullptr;       // tools: A/B switch back to the separate GELU passes
    if (off) return false;
    const int e = ab_dtype == DT_BF16 ? select_epi<BF16>(g) : ab_dtype == DT_F16 ? select_epi<F16>(g) : EPI_GENERIC;
    return !g.conv && !g.kmajor && g.map_mode != MAP_ROWNORM && (e == EPI_LIN16_GELU2 || e == EPI_LIN16_DGELU);
}

int launch_gemm(const GemmArgs& g_in, int ab_dtype, hipStream_t stream) {
    GemmArgs g = g_in;
    static const int dbg = getenv("LSEG_GEMM_DBG") ? atoi(getenv("LSEG_GEMM_DBG")) : 0;
    g.dbg = dbg;
    static const int group_env = getenv("LSEG_GEMM_GROUP_M") ? atoi(getenv("LSEG_GEMM_GROUP_M")) : 0;     // tools: A/B of the rasterisation group
    g.group_m = group_env > 0 ? group_env : 8;   // row-blocks per rasterisation group (2 .. 32 within noise in round 2 and again in round 6)
    static const int grid_cap = getenv("LSEG_GEMM_MAXGRID") ? atoi(getenv("LSEG_GEMM_MAXGRID")) : 0;     // tools: every launch on part of the chip
    if (grid_cap >= 8 && !g.max_grid) g.max_grid = grid_cap;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return set_error(LSEG_ERR_INVALID, "gemm: empty problem %dx%dx%d", g.M, g.N, g.K);
    if ((g.C_pre || g.dgelu_pre) && !gemm_fuses_gelu(g, ab_dtype))
        return set_error(LSEG_ERR_UNSUPPORTED, "gemm: C_pre / dgelu_pre need the specialised 16-bit MAP_LINEAR epilogue (N %% 128 == 0, a bias, 16-byte rows)");
    if (g.qkv_qscale != 0.f && !gemm_qkv_scales_q(g, ab_dtype))
        return set_error(LSEG_ERR_UNSUPPORTED, "gemm: qkv_qscale needs the specialised QKV epilogue");
    if (g.C_relu && !gemm_epilogue_is_pad16(g, ab_dtype)) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: C_relu needs the padded-NHWC specialised epilogue");
    if (g.K % 64 != 0) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: K=%d must be a multiple of 64", g.K);
    if (g.conv && (g.cin % 64 != 0)) return set_error(LSEG_ERR_UNSUPPORTED, "conv: Cin=%d must be a multiple of 64", g.cin);
    if (g.nsplit > 1 && (g.bias || g.res_mode != RES_NONE || g.map_mode != MAP_LINEAR || g.split || g.split_steps < 1 ||
                         (long)g.nsplit * g.split_steps < (g.K >> 6) || (long)(g.nsplit - 1) * g.split_steps >= (g.K >> 6)))
        return set_error(LSEG_ERR_INVALID, "split-K: needs a plain MAP_LINEAR GEMM without bias / residual and ranges that tile K exactly");
    if (g.split && g.relu_in) return set_error(LSEG_ERR_UNSUPPORTED, "split precision: ReLU on the A fragments is not representable (materialise relu(A))");
    if ((g.lda % 8) || (g.ldw % 8)) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: lda/ldw must be multiples of 8 elements");
    {   // per-lane source offsets are 32-bit byte offsets
        const double a_bytes = g.conv ? (double)g.M * g.stride * g.stride * 1.2 * g.cin * 2 + 4.0 * g.wp * g.cin * 2
                                      : (double)g.M * g.lda * 2;
        if (a_bytes >= 4.0e9 || (double)g.N * g.ldw * 2 >= 4.0e9)
            return set_error(LSEG_ERR_UNSUPPORTED, "gemm: operand larger than 4 GB (32-bit lane offsets)");
    }
    // the residual Linears of the ViT block: the hand-scheduled kernel (gemm_asm.hip) when asked for (LSEG_GEMM_ASM=1) and the buffers are tile-padded
    if (gemm_res32_asm_enabled() && gemm_res32_asm_eligible(g, ab_dtype)) return launch_gemm_res32_asm(g, ab_dtype, stream);
    if (ab_dtype == DT_BF16) return dispatch<BF16>(g, stream);
    if (ab_dtype == DT_F16) return dispatch<F16>(g, stream);
    return set_error(LSEG_ERR_INVALID, "gemm: operand dtype %d", ab_dtype);
}

}  // namespace lseg

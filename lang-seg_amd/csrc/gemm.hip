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
// Structure (cdna_hip_programming.md §5): BMxBNx64 tile, 4 waves (2x2), each wave a
// (BM/2)x(BN/2) sub-tile of v_mfma_f32_16x16x32 tiles; operands stream HBM -> LDS with
// direct-to-LDS loads (global_load_lds_dwordx4, 16 B/lane), two LDS stages, one barrier
// per K-step; XOR swizzle applied on the source address (LDS image stays lane-linear);
// XCD-aware tile rasterisation.  The MFMA is issued "swapped" (weights as the row
// operand) so each lane ends up with 4 consecutive output channels of one row ->
// 8/16-byte epilogue stores and float4 bias/residual loads.
#include "gemm.h"
#include "../../include/lseg_hip.h"

namespace lseg {

void gemm_args_init(GemmArgs& g) {
    g = GemmArgs{};
    g.stride = 1;
}

namespace {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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

template <typename T>
__device__ __attribute__((noinline)) void epilogue4(const GemmArgs& g, int m, int n, f32x4_t acc) {
    // 4 consecutive output columns n..n+3 of row m.
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    const int nvalid = (g.N - n) < 4 ? (g.N - n) : 4;
    if (nvalid <= 0) return;
    if (g.bias) {
        const int bn = g.bias_mod ? (n % g.bias_mod) : n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) v[r] += g.bias[bn + r];
    }
    if (g.round_mid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = to_f32<T>(from_f32<T>(v[r]));
    }
    if (g.act == ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
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

    // ---- destination offset -------------------------------------------------------------
    size_t off = 0;
    size_t estride = 1;   // element stride between the 4 consecutive n
    void* dst = g.C;
    switch (g.map_mode) {
        case MAP_LINEAR: off = (size_t)m * g.ldc + n; break;
        case MAP_PERIODIC: {
            const int q = m / g.p_div, r = m - q * g.p_div;
            off = (size_t)(q * g.p_mul + r + g.p_off) * g.ldc + n;
        } break;
        case MAP_PADDED: {
            const int hw = g.ho * g.wo;
            const int b = m / hw, p = m - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            off = ((size_t)(b * (g.ho + 2) + y + 1) * (g.wo + 2) + x + 1) * g.ldc + n;
        } break;
        case MAP_PIXSHUF: {
            const int hw = g.ho * g.wo;
            const int b = m / hw, p = m - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            const int ij = n / g.ps_C, co = n - ij * g.ps_C;
            const int i = ij / g.ps_s, j = ij - i * g.ps_s;
            const int Hd = g.ho * g.ps_s + 2, Wd = g.wo * g.ps_s + 2;
            off = ((size_t)(b * Hd + y * g.ps_s + i + 1) * Wd + x * g.ps_s + j + 1) * g.ps_C + co;
        } break;
        case MAP_QKV: {
            const int which = n / g.qkv_dim, rem = n - which * g.qkv_dim;
            const int head = rem >> 6, d = rem & 63;
            const int b = m / g.qkv_ntok, t = m - b * g.qkv_ntok;
            if (which == 2) {
                dst = g.Cv;
                off = ((size_t)(b * g.qkv_heads + head) * 64 + d) * g.qkv_npad + t;
                estride = g.qkv_npad;
            } else {
                dst = which == 0 ? g.C : g.Ck;
                off = ((size_t)(b * g.qkv_heads + head) * g.qkv_npad + t) * 64 + d;
            }
        } break;
        case MAP_NCHW: {
            const int b = m / g.p_div, p = m - b * g.p_div;
            off = ((size_t)b * g.N + n) * g.p_div + p;
            estride = g.p_div;
        } break;
    }

    // ---- residual(s) ----------------------------------------------------------------------
    if (g.res_mode != RES_NONE) {
        size_t roff = off;
        if (g.res_mode == RES_PERIODIC) roff = (size_t)((m % g.p_div) + g.p_off) * g.ldr + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) {
                v[r] += load_as_f32(g.res, roff + r * estride, g.res_dtype);
                if (g.res2) v[r] += load_as_f32(g.res2, roff + r * estride, g.res_dtype);
            }
    }

    // ---- store -------------------------------------------------------------------------------
    if (estride == 1 && nvalid == 4) {
        if (g.out_dtype == DT_F32) {
            *reinterpret_cast<float4*>((float*)dst + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint16_t h[4];
            if (g.out_dtype == DT_F16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = f32_to_f16(v[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = f32_to_bf16(v[r]);
            }
            uint2 pk;
            pk.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
            pk.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
            *reinterpret_cast<uint2*>((uint16_t*)dst + off) = pk;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) store_from_f32(dst, off + r * estride, g.out_dtype, v[r]);
    }
}

template <typename T, int BM, int BN, bool CONV, bool RELU_IN, int TAG>
__global__ __launch_bounds__(256) void lseg_gemm_kernel(const GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;     // per-wave sub-tile
    constexpr int MI = WM / 16, NI = WN / 16;   // 16x16 MFMA tiles per wave
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int A_SLABS = BM / 32, W_SLABS = BN / 32;   // 8-row slabs per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;

    // ---- XCD-aware rasterisation: consecutive tile ids (same A row-block) share an XCD/L2 ----
    const int tiles_n = (g.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int tile;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    // ---- per-lane source rows for the direct-to-LDS loads -----------------------------------
    const uint16_t* a_src[A_SLABS];
    const uint16_t* w_src[W_SLABS];
    int a_row[A_SLABS], w_row[W_SLABS];
#pragma unroll
    for (int s = 0; s < A_SLABS; ++s) {
        const int r = (s * 4 + w) * 8 + (lane >> 3);
        a_row[s] = r;
        int m = m0 + r;
        if (m > g.M - 1) m = g.M - 1;
        if (CONV) {
            const int hw = g.ho * g.wo;
            const int b = m / hw, p = m - b * hw;
            const int y = p / g.wo, x = p - y * g.wo;
            a_src[s] = g.A + ((size_t)(b * g.hp + y * g.stride) * g.wp + x * g.stride) * g.cin;
        } else {
            a_src[s] = g.A + (size_t)m * g.lda;
        }
    }
#pragma unroll
    for (int s = 0; s < W_SLABS; ++s) {
        const int r = (s * 4 + w) * 8 + (lane >> 3);
        w_row[s] = r;
        int n = n0 + r;
        if (n > g.N - 1) n = g.N - 1;
        w_src[s] = g.W + (size_t)n * g.ldw;
    }

    const int nk = g.K >> 6;
    const int cpt = CONV ? (g.cin >> 6) : 1;    // 64-wide K chunks per conv tap

    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE;
        char* sw = sa + A_BYTES;
        int koff_a;
        if (CONV) {
            const int tap = kt / cpt, ci0 = (kt - tap * cpt) << 6;
            const int ky = tap / 3, kx = tap - ky * 3;
            koff_a = (ky * g.wp + kx) * g.cin + ci0;
        } else {
            koff_a = kt << 6;
        }
        const int koff_w = kt << 6;
#pragma unroll
        for (int s = 0; s < A_SLABS; ++s)
            glds_slab_row(a_src[s] + koff_a, a_row[s], lane, sa + (s * 4 + w) * 1024);
#pragma unroll
        for (int s = 0; s < W_SLABS; ++s)
            glds_slab_row(w_src[s] + koff_w, w_row[s], lane, sw + (s * 4 + w) * 1024);
    };

    f32x4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0);   // own direct-to-LDS loads of tile kt have landed
        __syncthreads();                 // everyone's have; everyone is done reading stage (kt+1)&1
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + (lane >> 4);
            i32x4_t wf[NI], af[MI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = wn * WN + i * 16 + (lane & 15);
                wf[i] = *reinterpret_cast<const i32x4_t*>(sw + tile_off(r, c));
            }
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int r = wm * WM + j * 16 + (lane & 15);
                af[j] = *reinterpret_cast<const i32x4_t*>(sa + tile_off(r, c));
                if (RELU_IN) af[j] = relu_frag(af[j]);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
        }
    }

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + 0..3] ----------------
    // The (out-of-line) epilogue reads its parameters straight from the kernarg segment, so
    // the by-value GemmArgs is never copied to scratch.
    const GemmArgs& gk = *reinterpret_cast<const GemmArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m0 + wm * WM + j * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * WN + i * 16 + (lane >> 4) * 4;
            epilogue4<T>(gk, m, n, acc[i][j]);
        }
    }
}

template <typename T, int BM, int BN, bool CONV, bool RELU_IN, int TAG>
int launch_one(const GemmArgs& g, hipStream_t stream) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto kern = lseg_gemm_kernel<T, BM, BN, CONV, RELU_IN, TAG>;
    static bool attr_done = false;
    if (!attr_done) {
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, stream, g);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
int dispatch(const GemmArgs& g, hipStream_t stream) {
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const bool big = tiles128 >= 192;       // fills the 256 CUs reasonably with 128x128 tiles
    if (g.conv) {
        if (g.relu_in)
            return big ? launch_one<T, 128, 128, true, true, 0>(g, stream) : launch_one<T, 64, 64, true, true, 0>(g, stream);
        return big ? launch_one<T, 128, 128, true, false, 0>(g, stream) : launch_one<T, 64, 64, true, false, 0>(g, stream);
    }
    if (g.relu_in) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: relu_in is only implemented for the conv path");
    if (g.tag == 1) {
        return big ? launch_one<T, 128, 128, false, false, 1>(g, stream) : launch_one<T, 64, 64, false, false, 1>(g, stream);
    }
    return big ? launch_one<T, 128, 128, false, false, 0>(g, stream) : launch_one<T, 64, 64, false, false, 0>(g, stream);
}

}  // namespace

int launch_gemm(const GemmArgs& g, int ab_dtype, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return set_error(LSEG_ERR_INVALID, "gemm: empty problem %dx%dx%d", g.M, g.N, g.K);
    if (g.K % 64 != 0) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: K=%d must be a multiple of 64", g.K);
    if (g.conv && (g.cin % 64 != 0)) return set_error(LSEG_ERR_UNSUPPORTED, "conv: Cin=%d must be a multiple of 64", g.cin);
    if ((g.lda % 8) || (g.ldw % 8)) return set_error(LSEG_ERR_UNSUPPORTED, "gemm: lda/ldw must be multiples of 8 elements");
    if (ab_dtype == DT_BF16) return dispatch<BF16>(g, stream);
    if (ab_dtype == DT_F16) return dispatch<F16>(g, stream);
    return set_error(LSEG_ERR_INVALID, "gemm: operand dtype %d", ab_dtype);
}

}  // namespace lseg

// attention_bwd.hip -- backward of softmax(Q K^T * scale [+ causal mask]) V, head_dim 64, gfx950.  First (correctness-first) version of
// the attention brick of the training step (the reference gets it from torch autograd through
// [3P] timm Attention.forward, invoked at lseg_vit.py:196-197).
//
// Flash-style recomputation, nothing of size N x N touches HBM.  One workgroup = 4 waves owns one block of 64 KEYS of
// one (batch, head) and walks the query blocks:
//     S  = Q_i K_j^T                      P  = exp2(S * scale*log2e - L2_i)        (L2 = log2 of the row's softmax sum)
//     dV_j += P^T dO_i                    dP = dO_i V_j^T
//     dS = P o (dP - D_i) * scale         D_i = rowsum(dO_i o O_i)
//     dQ_i += dS K_j  (fp32 atomics)      dK_j += dS^T Q_i
// All five products run on v_mfma_f32_16x16x32 through one LDS-tile routine C[m][n] += sum_k A[m][k] B[n][k]: every
// operand is staged in LDS in the orientation that puts its contraction index on the fast axis (so Q, K, dO, P and dS
// are kept in both orientations).  Plain global loads / ds_write staging, 92 KB of LDS, one workgroup per CU -- the
// throughput work (direct-to-LDS rings, swizzled tiles, dQ without atomics) comes after parity.
//
// Layouts = the forward's: q,k [BH, Npad, 64]; vt [BH, 64, Npad]; o, dO [B, Ntok, H*64]; lse2 [BH, Npad] fp32.
// Outputs fp32: dq, dk, dv [BH, Npad, 64] (dq must be zeroed by the caller: it is accumulated with atomics).
#include "common.h"
#include "../../include/lseg_hip.h"

namespace lseg {

struct AttnBwdArgs {
    const uint16_t *q, *k, *vt, *o, *d_o;
    const float* lse2;
    float *dq, *dk, *dv;
    int B, H, ntok, npad, causal;
    float scale, scale_log2e;
};

namespace {

constexpr int LD = 72;      // LDS row stride in 16-bit elements (64 + 8: 144-byte rows keep 16-byte fragment reads aligned
                            // and spread consecutive rows over the banks)
typedef uint16_t Tile[64][LD];

// acc[i][j][r] += sum_k A[m][k] * Bm[n][k],  m = wm*32 + j*16 + (lane&15),  n = wn*32 + i*16 + (lane>>4)*4 + r
template <typename T>
__device__ __forceinline__ void tile_mma(const Tile& A, const Tile& Bm, f32x4_t (&acc)[2][2], int wm, int wn, int lane) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        i32x4_t bf[2], af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            bf[i] = *reinterpret_cast<const i32x4_t*>(&Bm[wn * 32 + i * 16 + (lane & 15)][ks * 32 + (lane >> 4) * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            af[j] = *reinterpret_cast<const i32x4_t*>(&A[wm * 32 + j * 16 + (lane & 15)][ks * 32 + (lane >> 4) * 8]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T>(bf[i], af[j], acc[i][j]);
    }
}

__device__ __forceinline__ void zero_acc(f32x4_t (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

template <typename T>
__global__ __launch_bounds__(256) void lseg_attention_bwd_kernel(const AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tile& sK = *reinterpret_cast<Tile*>(smem + 0 * sizeof(Tile));      // [key][d]
    Tile& sKt = *reinterpret_cast<Tile*>(smem + 1 * sizeof(Tile));     // [d][key]
    Tile& sV = *reinterpret_cast<Tile*>(smem + 2 * sizeof(Tile));      // [key][d]
    Tile& sQ = *reinterpret_cast<Tile*>(smem + 3 * sizeof(Tile));      // [q][d]
    Tile& sQt = *reinterpret_cast<Tile*>(smem + 4 * sizeof(Tile));     // [d][q]
    Tile& sdO = *reinterpret_cast<Tile*>(smem + 5 * sizeof(Tile));     // [q][d]
    Tile& sdOt = *reinterpret_cast<Tile*>(smem + 6 * sizeof(Tile));    // [d][q]
    Tile& sPt = *reinterpret_cast<Tile*>(smem + 7 * sizeof(Tile));     // [key][q]   (P is only consumed transposed)
    Tile& sdS = *reinterpret_cast<Tile*>(smem + 8 * sizeof(Tile));     // [q][key]
    Tile& sdSt = *reinterpret_cast<Tile*>(smem + 9 * sizeof(Tile));    // [key][q]
    float* sL = reinterpret_cast<float*>(smem + 10 * sizeof(Tile));    // [64] log2-sum-exp of the query rows
    float* sD = sL + 64;                                               // [64] rowsum(dO o O)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int jb = blockIdx.x, bh = blockIdx.y;
    const int b = bh / a.H, h = bh - b * a.H;
    const int k0 = jb * 64;
    const uint16_t* Qg = a.q + (size_t)bh * a.npad * 64;
    const uint16_t* Kg = a.k + (size_t)bh * a.npad * 64;
    const uint16_t* Vtg = a.vt + (size_t)bh * 64 * a.npad;
    const size_t orow = (size_t)a.H * 64;                              // row stride of o / dO

    // ---- this block's keys: K (both orientations) and V -------------------------------------------------------------
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        const uint16_t kv = Kg[(size_t)(k0 + r) * 64 + c];             // rows >= ntok are zero padding of the forward
        sK[r][c] = kv; sKt[c][r] = kv;
        sV[c][r] = Vtg[(size_t)r * a.npad + k0 + c];                   // vt[d = r][key = c] -> sV[key][d]
    }
    f32x4_t acc_dk[2][2], acc_dv[2][2];
    zero_acc(acc_dk); zero_acc(acc_dv);

    const int nqb = (a.ntok + 63) >> 6;
    for (int ib = 0; ib < nqb; ++ib) {
        const int q0 = ib * 64;
        __syncthreads();                                               // previous iteration's readers are done
        // ---- Q_i, dO_i (both orientations), L2_i, D_i ---------------------------------------------------------------
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const uint16_t qv = Qg[(size_t)(q0 + r) * 64 + c];
            sQ[r][c] = qv; sQt[c][r] = qv;
            uint16_t dv = 0;
            if (q0 + r < a.ntok) dv = a.d_o[((size_t)b * a.ntok + q0 + r) * orow + (size_t)h * 64 + c];
            sdO[r][c] = dv; sdOt[c][r] = dv;
        }
        {   // D[q] = sum_d dO[q][d] * O[q][d]: 4 threads per row, 16 elements each
            const int r = tid >> 2, part = tid & 3;
            float s = 0.f;
            if (q0 + r < a.ntok) {
                const size_t base = ((size_t)b * a.ntok + q0 + r) * orow + (size_t)h * 64 + part * 16;
#pragma unroll
                for (int e = 0; e < 16; ++e) s += to_f32<T>(a.d_o[base + e]) * to_f32<T>(a.o[base + e]);
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (part == 0) {
                sD[r] = s;
                sL[r] = q0 + r < a.ntok ? a.lse2[(size_t)bh * a.npad + q0 + r] : 0.f;
            }
        }
        __syncthreads();
        // ---- S = Q K^T, dP = dO V^T : lane holds [q = wm*32 + j*16 + (lane&15)][key = wn*32 + i*16 + (lane>>4)*4 + r] ----
        f32x4_t s_acc[2][2], p_acc[2][2];
        zero_acc(s_acc); zero_acc(p_acc);
        tile_mma<T>(sQ, sK, s_acc, wm, wn, lane);
        tile_mma<T>(sdO, sV, p_acc, wm, wn, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int qr = wm * 32 + j * 16 + (lane & 15);
                const float l2 = sL[qr], dd = sD[qr];
                const bool qok = q0 + qr < a.ntok;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kc = wn * 32 + i * 16 + (lane >> 4) * 4 + r;
                    const bool ok = qok && (k0 + kc < a.ntok) && (!a.causal || k0 + kc <= q0 + qr);   // CLIP's causal -inf mask
                    const float p = ok ? __builtin_amdgcn_exp2f(s_acc[i][j][r] * a.scale_log2e - l2) : 0.f;
                    const float ds = p * (p_acc[i][j][r] - dd) * a.scale;
                    const uint16_t pb = from_f32<T>(p), dsb = from_f32<T>(ds);
                    sPt[kc][qr] = pb;
                    sdS[qr][kc] = dsb; sdSt[kc][qr] = dsb;
                }
            }
        __syncthreads();
        // ---- dV_j += P^T dO_i ; dK_j += dS^T Q_i : lane holds [key = wm*32 + ..][d = wn*32 + ..] ------------------------
        tile_mma<T>(sPt, sdOt, acc_dv, wm, wn, lane);
        tile_mma<T>(sdSt, sQt, acc_dk, wm, wn, lane);
        // ---- dQ_i += dS K_j : lane holds [q][d], accumulated over the key blocks with fp32 atomics ----------------------
        f32x4_t q_acc[2][2];
        zero_acc(q_acc);
        tile_mma<T>(sdS, sKt, q_acc, wm, wn, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int qr = q0 + wm * 32 + j * 16 + (lane & 15);
                if (qr >= a.ntok) continue;
                float* dst = a.dq + ((size_t)bh * a.npad + qr) * 64 + wn * 32 + i * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dst + r, q_acc[i][j][r]);
            }
    }
    // ---- dK_j, dV_j [key][d] ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kr = k0 + wm * 32 + j * 16 + (lane & 15);
            if (kr >= a.npad) continue;
            const size_t off = ((size_t)bh * a.npad + kr) * 64 + wn * 32 + i * 16 + (lane >> 4) * 4;
            *reinterpret_cast<float4*>(a.dk + off) = make_float4(acc_dk[i][j][0], acc_dk[i][j][1], acc_dk[i][j][2], acc_dk[i][j][3]);
            *reinterpret_cast<float4*>(a.dv + off) = make_float4(acc_dv[i][j][0], acc_dv[i][j][1], acc_dv[i][j][2], acc_dv[i][j][3]);
        }
}

}  // namespace

int launch_attention_backward(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse2,
                              float* dq, float* dk, float* dv, int B, int H, int ntok, int npad, int dtype, int causal, float scale,
                              hipStream_t stream) {
    if (npad % 64 != 0 || npad < ntok) return set_error(LSEG_ERR_INVALID, "attention backward: npad=%d must be a multiple of 64 and >= ntok=%d", npad, ntok);
    AttnBwdArgs a;
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.vt = (const uint16_t*)vt; a.o = (const uint16_t*)o; a.d_o = (const uint16_t*)d_o;
    a.lse2 = lse2; a.dq = dq; a.dk = dk; a.dv = dv;
    a.B = B; a.H = H; a.ntok = ntok; a.npad = npad; a.causal = causal; a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    const size_t lds = 10 * sizeof(Tile) + 2 * 64 * sizeof(float);
    dim3 grid((ntok + 63) / 64, B * H);
    if (dtype == DT_BF16) {          // the dynamic-LDS opt-in is per device: set it on every launch (microseconds; correctness-first kernel)
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lseg_attention_bwd_kernel<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lseg_attention_bwd_kernel<BF16>, grid, dim3(256), lds, stream, a);
    } else if (dtype == DT_F16) {
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lseg_attention_bwd_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lseg_attention_bwd_kernel<F16>, grid, dim3(256), lds, stream, a);
    } else {
        return set_error(LSEG_ERR_INVALID, "attention backward: dtype %d", dtype);
    }
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace lseg

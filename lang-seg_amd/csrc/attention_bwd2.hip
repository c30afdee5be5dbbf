// attention_bwd2.hip -- backward of softmax(Q K^T * scale) V (head_dim 64, no mask) as the training step runs it
// ([3P] timm Attention under autograd, invoked from lseg_vit.py:196-197; modules/lsegmentation_module.py:66-81).
//
// Flash-style recomputation from the forward's saved log-sum-exp, in the forward kernel's structure (attention.hip): MFMA 32x32x16 in the
// "transposed" orientation so that the softmax index a lane needs per-row scalars for lives on the lane, operand tiles streamed HBM -> LDS
// by direct-to-LDS loads into swizzled 128-byte rows, exponentiated scores regrouped into operand order with v_permlane32_swap.
// No atomics: two kernels, each owning its outputs --
//   dq kernel : one workgroup = 128 queries of one (batch, head), walks the key tiles          (3 products per tile: S^T, dP^T, dQ^T)
//   dkv kernel: one workgroup = 128 keys,                        walks the query tiles         (4 products per tile: S, dP, dV^T, dK^T)
// and both write bf16 straight into d(qkv Linear output) [B*Ntok, 3*H*64], the dY operand of the qkv layer's backward GEMMs.
// Every streamed tile must be [rows][64 contiguous]: a prep kernel lays out, per layer, K^T, Q^T, V (row-major), dO head-major, dO^T and
// D = rowsum(dO o O) from the forward's q, k, v^T, O and the incoming dO (token-major).
//
//   S = Q K^T ; P = exp2(S * scale*log2e - L2) ; dP = dO V^T ; dS = P o (dP - D) * scale ; dQ = dS K ; dK = dS^T Q ; dV = P^T dO
#include "ops.h"
#include "../../include/lseg_hip.h"

namespace lseg {

struct AttnBwd2Args {
    const uint16_t *q, *k, *v, *qT, *kT, *dO, *dOT;     // q,k,v,dO: [BH, npad, 64]; qT,kT,dOT: [BH, 64, npad]
    const float *lse2, *dsum;                           // [BH, npad]
    uint16_t* dqkv;                                     // [B*ntok, 3*H*64]
    int B, H, ntok, npad;
    float scale, scale_log2e;
};

namespace {

// q,k [BH,npad,64], vt [BH,64,npad], o / dO token-major [B,ntok,H*64]  ->  qT, kT [BH,64,npad]; v, dO_hm [BH,npad,64]; dOT [BH,64,npad];
// dsum [BH,npad].  One workgroup per (64-token block, bh); 64x64 tiles through LDS, 16 bytes per lane on every global access (with 2-byte
// accesses the kernel ran at 3 TB/s: 50 us of the 150 MB it moves per layer at B = 8).  LDS rows are 66 elements: the transposing reads
// (8 two-byte reads down a column per 16-byte store) are bank-conflict free, the row writes go in as 4-byte words.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ vt,
                                                            const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o, uint16_t* __restrict__ qT,
                                                            uint16_t* __restrict__ kT, uint16_t* __restrict__ v, uint16_t* __restrict__ dohm,
                                                            uint16_t* __restrict__ doT, float* __restrict__ dsum, int B, int H, int ntok, int npad) {
    __shared__ uint16_t t0[64][66], t1[64][66], t2[64][66], t3[64][66];
    const int bh = blockIdx.y, r0 = blockIdx.x * 64, b = bh / H, h = bh - b * H;
    const size_t rs = (size_t)H * 64;
    auto put = [](uint16_t (&t)[64][66], int r, int c, const uint4& u) {
        uint32_t* d = reinterpret_cast<uint32_t*>(&t[r][c]);
        d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w;
    };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = threadIdx.x + it * 256, r = i >> 3, c = (i & 7) * 8;          // 16-byte chunk (r, c..c+7) of a 64 x 64 tile
        const uint4 uq = *reinterpret_cast<const uint4*>(q + ((size_t)bh * npad + r0 + r) * 64 + c);
        const uint4 uk = *reinterpret_cast<const uint4*>(k + ((size_t)bh * npad + r0 + r) * 64 + c);
        const uint4 uv = *reinterpret_cast<const uint4*>(vt + ((size_t)bh * 64 + r) * npad + r0 + c);      // [d = r][key = c]
        uint4 ug = make_uint4(0u, 0u, 0u, 0u), uo = ug;
        if (r0 + r < ntok) {
            const size_t base = ((size_t)b * ntok + r0 + r) * rs + (size_t)h * 64 + c;
            ug = *reinterpret_cast<const uint4*>(d_o + base);
            uo = *reinterpret_cast<const uint4*>(o + base);
        }
        put(t0, r, c, uq); put(t1, r, c, uk); put(t2, r, c, uv); put(t3, r, c, ug);
        *reinterpret_cast<uint4*>(dohm + ((size_t)bh * npad + r0 + r) * 64 + c) = ug;
        // D[t] = sum_d dO[t,d] * O[t,d]: the 8 lanes holding a row
        const uint16_t *eg = reinterpret_cast<const uint16_t*>(&ug), *eo = reinterpret_cast<const uint16_t*>(&uo);
        float sd = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sd += to_f32<T>(eg[e]) * to_f32<T>(eo[e]);
        sd += __shfl_xor(sd, 1);
        sd += __shfl_xor(sd, 2);
        sd += __shfl_xor(sd, 4);
        if ((i & 7) == 0) dsum[(size_t)bh * npad + r0 + r] = sd;
    }
    __syncthreads();
    auto col = [](const uint16_t (&t)[64][66], int c0, int r) {                     // t[c0 .. c0+7][r] as one 16-byte vector
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = t[c0 + j][r];
        return make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
    };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = threadIdx.x + it * 256, r = i >> 3, c = (i & 7) * 8;          // output row r (= d or key), columns c..c+7
        *reinterpret_cast<uint4*>(qT + ((size_t)bh * 64 + r) * npad + r0 + c) = col(t0, c, r);
        *reinterpret_cast<uint4*>(kT + ((size_t)bh * 64 + r) * npad + r0 + c) = col(t1, c, r);
        *reinterpret_cast<uint4*>(doT + ((size_t)bh * 64 + r) * npad + r0 + c) = col(t3, c, r);
        *reinterpret_cast<uint4*>(v + ((size_t)bh * npad + r0 + r) * 64 + c) = col(t2, c, r);                // [key = r][d = c]
    }
}

// the accumulator of a transposed-orientation product, packed into B-operand fragments (8 contiguous contraction indices per lane):
// see attention.hip ("P^T fragments")
template <typename T>
__device__ __forceinline__ void to_operand(const f32x16_t (&s)[2], i32x4_t (&pf)[2][2]) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint32_t a0 = pack2<T>(s[sub][8 * s2 + 0], s[sub][8 * s2 + 1]);
            const uint32_t a1 = pack2<T>(s[sub][8 * s2 + 2], s[sub][8 * s2 + 3]);
            const uint32_t b0 = pack2<T>(s[sub][8 * s2 + 4], s[sub][8 * s2 + 5]);
            const uint32_t b1 = pack2<T>(s[sub][8 * s2 + 6], s[sub][8 * s2 + 7]);
            const auto w0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto w1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            pf[sub][s2][0] = (int)w0[0]; pf[sub][s2][1] = (int)w1[0];
            pf[sub][s2][2] = (int)w0[1]; pf[sub][s2][3] = (int)w1[1];
        }
}

// out^T accumulators (lane = token of this wave, registers = the 64 d) -> bf16 row of d(qkv): 4 consecutive d per 8-byte store
template <typename T>
__device__ __forceinline__ void store_rows(const f32x16_t (&o)[2], uint16_t* row, int hi) {
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack2<T>(o[d][g * 4 + 0], o[d][g * 4 + 1]);
            pk.y = pack2<T>(o[d][g * 4 + 2], o[d][g * 4 + 3]);
            *reinterpret_cast<uint2*>(row + d * 32 + g * 8 + hi * 4) = pk;
        }
}

// XCD-aware block -> (row block, head) map (as in attention.hip): all row blocks of a head run on one XCD, back to back, so the head's
// streamed operands are fetched into ONE private L2 instead of all eight.  1-D grid of nb * BH workgroups.
__device__ __forceinline__ void xcd_head_map(int nb, int BH, int& blk, int& bh) {
    const int L = blockIdx.x;
    blk = L % nb; bh = L / nb;
    if ((BH & 7) == 0) {
        const int xcd = L & 7, idx = L >> 3;
        blk = idx % nb;
        bh = (idx / nb) * 8 + xcd;
    }
}

// ---- dQ: lane = query (per-lane L2 and D), key tiles stream through LDS: K, V [64 keys][64 d] and K^T [64 d][64 keys] ------------------
template <typename T>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const AttnBwd2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (K 8 KB + V 8 KB + K^T 8 KB)
    constexpr int TILE = 8192, STAGE = 3 * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ql = lane & 31;
    int blk, bh;
    xcd_head_map((a.ntok + 127) / 128, a.B * a.H, blk, bh);
    const int q0 = (blk * 4 + w) * 32;
    const size_t hb = (size_t)bh * a.npad * 64;
    int qr = q0 + ql;
    if (qr > a.npad - 1) qr = a.npad - 1;
    i32x4_t qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const i32x4_t*>(a.q + hb + (size_t)qr * 64 + ks * 16 + hi * 8);
        dof[ks] = *reinterpret_cast<const i32x4_t*>(a.dO + hb + (size_t)qr * 64 + ks * 16 + hi * 8);
    }
    const float L = a.lse2[(size_t)bh * a.npad + qr], Dq = a.dsum[(size_t)bh * a.npad + qr];
    const int n_tiles = (a.ntok + 63) >> 6;
    uint32_t r_off[2], t_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = (s * 4 + w) * 8 + (lane >> 3);
        const int ch = ((lane & 7) ^ swz(r)) << 3;
        r_off[s] = (uint32_t)(r * 64 + ch) * 2u;            // row-major [token][64] tiles
        t_off[s] = (uint32_t)(r * a.npad + ch) * 2u;        // transposed [d][npad] tiles
    }
    auto issue = [&](int t, int stage) {
        char* sk = smem + stage * STAGE;
        // SGPR bases + 32-bit lane offsets (see attention.hip: keeps the loads on the saddr form, no 64-bit VALU adds per tile)
        const char* kb = uniform_ptr(reinterpret_cast<const char*>(a.k + hb + (size_t)t * 64 * 64));
        const char* vb = uniform_ptr(reinterpret_cast<const char*>(a.v + hb + (size_t)t * 64 * 64));
        const char* tb = uniform_ptr(reinterpret_cast<const char*>(a.kT + hb + (size_t)t * 64));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t ro = r_off[s], to = t_off[s];
            asm volatile("" : "+v"(ro), "+v"(to));
            glds_slab_off(kb, ro, sk + (s * 4 + w) * 1024);
            glds_slab_off(vb, ro, sk + TILE + (s * 4 + w) * 1024);
            glds_slab_off(tb, to, sk + 2 * TILE + (s * 4 + w) * 1024);
        }
    };
    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    const bool live = q0 < a.ntok;
    issue(0, 0);
    for (int t = 0; t < n_tiles; ++t) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (t + 1 < n_tiles) issue(t + 1, (t + 1) & 1);
        if (!live) continue;
        const char* sk = smem + (t & 1) * STAGE;
        const char* sv = sk + TILE;
        const char* st = sk + 2 * TILE;
        f32x16_t s[2], dp[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[sub][r] = 0.f; dp[sub][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const i32x4_t kf = *reinterpret_cast<const i32x4_t*>(sk + tile_off(sub * 32 + ql, ks * 2 + hi));
                const i32x4_t vf = *reinterpret_cast<const i32x4_t*>(sv + tile_off(sub * 32 + ql, ks * 2 + hi));
                s[sub] = mfma32<T>(kf, qf[ks], s[sub]);          // S^T[key][q]
                dp[sub] = mfma32<T>(vf, dof[ks], dp[sub]);       // dP^T[key][q]
            }
        }
        const bool tail = t * 64 + 64 > a.ntok;                  // wave-uniform: padded keys must not contribute
        // dS^T = P^T o (dP^T - D); the softmax scale is applied once to dQ at the end (4 VALU per score: fma, exp2, sub, mul -- plain
        // fp32 forms: the packed ones run at half rate on gfx950 and do not overlap with other waves' MFMAs, tools/probes/valu_mfma_probe.py)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = __builtin_amdgcn_exp2f(fmaf(s[sub][r], a.scale_log2e, -L));      // P^T
        if (tail) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s[sub][r] = key < a.ntok ? s[sub][r] : 0.f;
                }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] *= dp[sub][r] - Dq;
        i32x4_t dsf[2][2];
        to_operand<T>(s, dsf);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int row = d * 32 + ql;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const i32x4_t tf = *reinterpret_cast<const i32x4_t*>(st + tile_off(row, sub * 4 + s2 * 2 + hi));
                    o[d] = mfma32<T>(tf, dsf[sub][s2], o[d]);    // dQ^T[d][q] += K^T[d][keys] dS^T[keys][q]
                }
        }
    }
    const int qrow = q0 + ql;
    if (qrow < a.ntok) {
        const int b = bh / a.H, h = bh - b * a.H;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= a.scale;
        store_rows<T>(o, a.dqkv + ((size_t)b * a.ntok + qrow) * (3 * a.H * 64) + h * 64, hi);
    }
}

// ---- dK, dV: lane = key; query tiles stream through LDS: Q, dO [64 q][64 d], Q^T, dO^T [64 d][64 q], L2 and D of the 64 queries --------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnBwd2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (4 x 8 KB + 512 B)
    constexpr int TILE = 8192, STAGE = 4 * TILE + 512;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ql = lane & 31;
    int blk, bh;
    xcd_head_map((a.ntok + 127) / 128, a.B * a.H, blk, bh);
    const int k0 = (blk * 4 + w) * 32;
    const size_t hb = (size_t)bh * a.npad * 64;
    int kr = k0 + ql;
    if (kr > a.npad - 1) kr = a.npad - 1;
    i32x4_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *reinterpret_cast<const i32x4_t*>(a.k + hb + (size_t)kr * 64 + ks * 16 + hi * 8);
        vf[ks] = *reinterpret_cast<const i32x4_t*>(a.v + hb + (size_t)kr * 64 + ks * 16 + hi * 8);
    }
    const int n_tiles = (a.ntok + 63) >> 6;
    uint32_t r_off[2], t_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = (s * 4 + w) * 8 + (lane >> 3);
        const int ch = ((lane & 7) ^ swz(r)) << 3;
        r_off[s] = (uint32_t)(r * 64 + ch) * 2u;
        t_off[s] = (uint32_t)(r * a.npad + ch) * 2u;
    }
    auto issue = [&](int t, int stage) {
        char* sq = smem + stage * STAGE;
        const char* qb = uniform_ptr(reinterpret_cast<const char*>(a.q + hb + (size_t)t * 64 * 64));
        const char* ob = uniform_ptr(reinterpret_cast<const char*>(a.dO + hb + (size_t)t * 64 * 64));
        const char* qtb = uniform_ptr(reinterpret_cast<const char*>(a.qT + hb + (size_t)t * 64));
        const char* otb = uniform_ptr(reinterpret_cast<const char*>(a.dOT + hb + (size_t)t * 64));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t ro = r_off[s], to = t_off[s];
            asm volatile("" : "+v"(ro), "+v"(to));
            glds_slab_off(qb, ro, sq + (s * 4 + w) * 1024);
            glds_slab_off(ob, ro, sq + TILE + (s * 4 + w) * 1024);
            glds_slab_off(qtb, to, sq + 2 * TILE + (s * 4 + w) * 1024);
            glds_slab_off(otb, to, sq + 3 * TILE + (s * 4 + w) * 1024);
        }
        if (w == 0) {      // L2 and D of this tile's 64 queries: 4 bytes per lane, direct to LDS (lane-linear)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.lse2 + (size_t)bh * a.npad + t * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(sq + 4 * TILE), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.dsum + (size_t)bh * a.npad + t * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(sq + 4 * TILE + 256), 4, 0, 0);
        }
    };
    f32x16_t dv[2], dk[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[d][r] = 0.f; dk[d][r] = 0.f; }
    const bool live = k0 < a.ntok;
    issue(0, 0);
    for (int t = 0; t < n_tiles; ++t) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (t + 1 < n_tiles) issue(t + 1, (t + 1) & 1);
        if (!live) continue;
        const char* sq = smem + (t & 1) * STAGE;
        const char* so = sq + TILE;
        const char* sqt = sq + 2 * TILE;
        const char* sot = sq + 3 * TILE;
        const float* sL = reinterpret_cast<const float*>(sq + 4 * TILE);
        const float* sD = sL + 64;
        f32x16_t s[2], dp[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[sub][r] = 0.f; dp[sub][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const i32x4_t qa = *reinterpret_cast<const i32x4_t*>(sq + tile_off(sub * 32 + ql, ks * 2 + hi));
                const i32x4_t oa = *reinterpret_cast<const i32x4_t*>(so + tile_off(sub * 32 + ql, ks * 2 + hi));
                s[sub] = mfma32<T>(qa, kf[ks], s[sub]);          // S[q][key]
                dp[sub] = mfma32<T>(oa, vf[ks], dp[sub]);        // dP[q][key]
            }
        }
        f32x16_t ds[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // registers 4g..4g+3 hold queries sub*32 + 8g + 4hi + (0..3): one 16-byte LDS read each for L2 and D
                const float4 l4 = *reinterpret_cast<const float4*>(sL + sub * 32 + 8 * g + 4 * hi);
                const float4 d4 = *reinterpret_cast<const float4*>(sD + sub * 32 + 8 * g + 4 * hi);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[sub][r], a.scale_log2e, -lv[e]));
                    s[sub][r] = p;                               // P (rows beyond ntok: dO = 0 and D = 0 there, they add nothing)
                    ds[sub][r] = p * (dp[sub][r] - dvv[e]);      // dS without the softmax scale: applied once to dK at the end
                }
            }
        i32x4_t pf[2][2], dsf[2][2];
        to_operand<T>(s, pf);
        to_operand<T>(ds, dsf);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int row = d * 32 + ql;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const i32x4_t otf = *reinterpret_cast<const i32x4_t*>(sot + tile_off(row, sub * 4 + s2 * 2 + hi));
                    const i32x4_t qtf = *reinterpret_cast<const i32x4_t*>(sqt + tile_off(row, sub * 4 + s2 * 2 + hi));
                    dv[d] = mfma32<T>(otf, pf[sub][s2], dv[d]);      // dV^T[d][key] += dO^T[d][q] P[q][key]
                    dk[d] = mfma32<T>(qtf, dsf[sub][s2], dk[d]);     // dK^T[d][key] += Q^T[d][q] dS[q][key]
                }
        }
    }
    const int krow = k0 + ql;
    if (krow < a.ntok) {
        const int b = bh / a.H, h = bh - b * a.H, D = a.H * 64;
        uint16_t* row = a.dqkv + ((size_t)b * a.ntok + krow) * (3 * D) + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[d][r] *= a.scale;
        store_rows<T>(dk, row + D, hi);
        store_rows<T>(dv, row + 2 * D, hi);
    }
}

}  // namespace

size_t attention_backward_ws_bytes(int B, int H, int npad) {
    const size_t plane = (size_t)B * H * npad * 64 * 2;
    return 5 * plane + (size_t)B * H * npad * sizeof(float);
}

// d(qkv) [B*ntok, 3*H*64] (bf16/fp16) from q, k [BH,npad,64], vt [BH,64,npad], o / d_o [B,ntok,H*64] and the forward's lse2 [BH,npad];
// ws: attention_backward_ws_bytes(B, H, npad) bytes of scratch.
int launch_attention_backward_qkv(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse2, void* dqkv,
                                  void* ws, int B, int H, int ntok, int npad, int dtype, float scale, hipStream_t stream) {
    if (npad % 128 != 0 || npad < ntok) return set_error(LSEG_ERR_INVALID, "attention backward: npad=%d must be a multiple of 128 and >= ntok=%d", npad, ntok);
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)ws) & 15)
        return set_error(LSEG_ERR_INVALID, "attention backward: operands must be 16-byte aligned");
    const size_t pe = (size_t)B * H * npad * 64;
    uint16_t* base = (uint16_t*)ws;
    uint16_t *qT = base, *kT = base + pe, *v = base + 2 * pe, *dohm = base + 3 * pe, *doT = base + 4 * pe;
    float* dsum = (float*)(base + 5 * pe);
    AttnBwd2Args a;
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = v; a.qT = qT; a.kT = kT; a.dO = dohm; a.dOT = doT;
    a.lse2 = lse2; a.dsum = dsum; a.dqkv = (uint16_t*)dqkv;
    a.B = B; a.H = H; a.ntok = ntok; a.npad = npad; a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    dim3 gp(npad / 64, B * H), g(((ntok + 127) / 128) * B * H);
    const size_t lds_dq = 2 * 3 * 8192, lds_dkv = 2 * (4 * 8192 + 512);
#define RUN(TT)                                                                                                                              \
    do {                                                                                                                                     \
        hipLaunchKernelGGL(attn_bwd_prep_kernel<TT>, gp, dim3(256), 0, stream, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)vt,   \
                           (const uint16_t*)o, (const uint16_t*)d_o, qT, kT, v, dohm, doT, dsum, B, H, ntok, npad);                          \
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds_dkv));                                                                                     \
        hipLaunchKernelGGL(attn_bwd_dq_kernel<TT>, g, dim3(256), lds_dq, stream, a);                                                          \
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<TT>, g, dim3(256), lds_dkv, stream, a);                                                        \
    } while (0)
    if (dtype == DT_BF16) RUN(BF16);
    else if (dtype == DT_F16) RUN(F16);
    else return set_error(LSEG_ERR_INVALID, "attention backward: dtype %d", dtype);
#undef RUN
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace lseg

// attention.hip -- fused softmax(Q K^T * scale [+ causal mask]) V for head_dim 64 on gfx950.
//
// Replaces the materialised attention of
//   [3P] timm 0.4.12 Attention.forward (invoked from lseg_vit.py:196-197; restated by the
//        reference's own hook at lseg_vit.py:24-40): 901 tokens, 16 heads, fp32 scores
//   [3P] CLIP nn.MultiheadAttention with the causal -inf mask (clip/model.py), 77 tokens
// with a flash-style single pass: scores never touch HBM.
//
// Layout: q,k [BH, Npad, 64]; v TRANSPOSED vt [BH, 64, Npad] (the QKV GEMM epilogue writes
// these directly); out [B, Ntok, H*64] row-major = the A operand of the projection GEMM.
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows.
// K / V^T tiles of 64 keys stream HBM -> LDS with direct-to-LDS loads (double-buffered).
//
// Both products use v_mfma_f32_32x32x16 in the "transposed" orientation so that the query
// index lives on the lane (col = lane&31) for S^T = K Q^T AND for O^T = V^T P^T:
//   * row max / row sum are lane-local (plus one exchange with lane^32),
//   * the online-softmax rescale of O is a per-lane scalar multiply,
//   * the exponentiated scores feed the second MFMA straight from registers: the k-slot
//     accumulator layout of the first (key = 16*s + 4*(lane>>5) + (j&3) + 8*(j>>2)) is turned
//     into the standard operand order (8 contiguous keys per lane) by one v_permlane32_swap per
//     packed word -- no LDS round trip of P -- and V^T is read with conflict-free ds_read_b128.
#include <cstdlib>
#include <type_traits>

#include "ops.h"
#include "../../include/lseg_hip.h"

namespace lseg {

struct AttnArgs {
    const uint16_t* q; const uint16_t* k; const uint16_t* vt;
    uint16_t* out;
    float* lse2;           // optional [BH, npad] fp32: log2 of each query row's sum_k exp2(s_k * scale * log2e) (saved for the backward)
    int B, H, ntok, npad, causal;
    float scale_log2e;     // softmax scale * log2(e)
    int prescaled;         // q already carries scale * log2(e) (the QKV epilogue folded it into its single rounding): S^T comes out of the
                           // MFMA in exp2 units
};

namespace {

// phase-probe hooks (tools/probes/attn_phase_probe.py compiles this file with them defined; empty in the library build)
#ifndef ATTN_PROBE
#define ATTN_PROBE(i)
#define ATTN_PROBE_DECL
#define ATTN_PROBE_DUMP
#endif

// NW waves per workgroup = 32 * NW query rows: 4 for the bulk shapes; 2 when the grid would not fill the chip (B = 1: 8 x 16 workgroups of
// 4 waves on 256 CUs -> 15 x 16 of 2)
// NS K / V^T stages in LDS: tile t + NS - 1 is requested while tile t is computed (NS = 3: a direct-to-LDS load has two tile times to land)
// Tile body (DESIGN.md par. 3.2): all 8 K fragments of a tile are requested before the first QK^T MFMA and the 8 V^T fragments right
// after the last one (they land under the softmax VALU); the two accumulators of each product alternate, so no MFMA waits for an LDS
// round trip or for its predecessor's result.  (Round 3's body had every MFMA behind its own ds_read + lgkmcnt(0); same arithmetic.)
// PRE = false: q as the Linear wrote it, softmax scale applied to the scores (training forward -- the backward consumes q --, the causal
//   text tower, the C-ABI brick lseg_op_attention).  Bit-identical to round 3's kernel.
// PRE = true (the inference image tower): q PRE-SCALED by scale * log2(e) in the QKV epilogue's single rounding, and the REFERENCE MAX
//   BAKED INTO THE ACCUMULATOR: S^T starts from -m_ref (a per-query constant = a per-lane constant in this orientation), so the matrix
//   pipe delivers the exp2 argument itself -- no fma per score.  m_ref is the running max, refreshed only when a tile's relative max
//   exceeds 2^THR (exact: the rescale of O and l is the usual one; every p <= 2^THR fits the 16-bit operand type and fp32 sums); the
//   first tile always takes the refresh path.  The row sums come from the matrix pipe too: l += 1^T P^T as four more MFMAs per tile on
//   a ones operand (they sum exactly the ROUNDED probabilities the PV product uses) instead of 32 v_add_f32.
//   Measured at B = 36 (tools/attention_bench.py, profiles/r04_attention_experiments.txt): bf16 221.6 -> 196.5 us, fp16 225.0 -> 212.9.
template <typename T, int NW, int NS, bool PRE>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 3) void lseg_attention_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NS x (K 8 KB + Vt 8 KB)
    constexpr int TILE = 8192, STAGE = 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform
    const int hi = lane >> 5, ql = lane & 31;
    // XCD-aware block -> (query block, head) map.  Workgroups are dealt to the 8 XCDs round-robin in linear order (x fastest), so with
    // the plain map XCD c would run query block c of EVERY head and each of the 8 private L2s would stream every head's K / V^T
    // (measured: L2 hit rate 27 %, 1.27 GB of fabric traffic per launch = 6.3 TB/s at B = 36 -- the kernel was fabric-bound).  Here all
    // query blocks of a head run on ONE XCD, back to back: its K / V^T (256 KB) is fetched once into that XCD's 4 MB L2.
    const int nqb = (a.ntok + 32 * NW - 1) / (32 * NW), BH = a.B * a.H;       // 1-D grid of nqb * BH workgroups
    const int L = blockIdx.x;
    int qb = L % nqb, bh = L / nqb;
    if ((BH & 7) == 0) {
        const int xcd = L & 7, idx = L >> 3;
        qb = idx % nqb;
        bh = (idx / nqb) * 8 + xcd;
    }
    const int q0 = (qb * NW + w) * 32;
    const uint16_t* Q = a.q + (size_t)bh * a.npad * 64;
    const uint16_t* K = a.k + (size_t)bh * a.npad * 64;
    const uint16_t* Vt = a.vt + (size_t)bh * 64 * a.npad;

    // Q fragments: B operand of S^T = K Q^T; lane -> col q, k-slots d = ks*16 + hi*8 + j
    i32x4_t qf[4];
    {
        int qr = q0 + ql;
        if (qr > a.npad - 1) qr = a.npad - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const i32x4_t*>(Q + (size_t)qr * 64 + ks * 16 + hi * 8);
    }

    int kv_end = a.ntok;
    if (a.causal) {
        const int qend = (qb + 1) * 32 * NW;
        kv_end = qend < a.ntok ? qend : a.ntok;
    }
    const int n_tiles = (kv_end + 63) >> 6;

    // per-lane byte offsets of the two K rows / two V^T rows this lane streams per tile (swizzled chunk)
    constexpr int SPW = 8 / NW;                // 8-row slabs of a 64-row tile per wave
    uint32_t k_off[SPW], v_off[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int r = (s * NW + w) * 8 + (lane >> 3);
        const int ch = ((lane & 7) ^ swz(r)) << 3;
        k_off[s] = (uint32_t)(r * 64 + ch) * 2u;
        v_off[s] = (uint32_t)(r * a.npad + ch) * 2u;
    }
    auto issue = [&](int t, int stage) {
        char* sk = smem + stage * STAGE;
        char* sv = sk + TILE;
        // wave-uniform bases, pinned to SGPRs: otherwise the loop is strength-reduced into per-lane 64-bit pointers and every
        // direct-to-LDS load of the loop costs a v_lshl_add_u64 pair (16 double-rate VALU instructions per tile) instead of the
        // SGPR-base + 32-bit VGPR-offset form
        const char* kb = uniform_ptr(reinterpret_cast<const char*>(K + (size_t)t * 64 * 64));
        const char* vb = uniform_ptr(reinterpret_cast<const char*>(Vt + (size_t)t * 64));
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            uint32_t ko = k_off[s], vo = v_off[s];
            asm volatile("" : "+v"(ko), "+v"(vo));         // keeps the 32-bit offsets from being hoisted as zero-extended 64-bit pairs
            glds_slab_off(kb, ko, sk + (s * NW + w) * 1024);
            glds_slab_off(vb, vo, sv + (s * NW + w) * 1024);
        }
    };

    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = PRE ? 0.f : -1e30f, l_run = 0.f;
    f32x16_t lacc;                            // PRE: every register = this query's running sum (1^T P^T on the matrix pipe)
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
    const int one2 = std::is_same<T, F16>::value ? 0x3C003C00 : 0x3F803F80;       // two 16-bit ones
    const i32x4_t ones = {one2, one2, one2, one2};
    const int qrow = q0 + ql;
    const bool live = q0 < a.ntok;      // wave-uniform (ntok = 901: 3 of the last block's 4 waves are padding only)

    issue(0, 0);
    if (NS == 3 && n_tiles > 1) issue(1, 1);
    ATTN_PROBE_DECL
    int st_cur = 0, st_nxt = NS - 1;      // stage of tile t; stage the newly requested tile t + NS - 1 goes to
    for (int t = 0; t < n_tiles; ++t) {
        // tile t has landed: with three stages the loads of tile t + 1 (the 2 * SPW most recent of this wave) may stay in flight
        if (NS == 3 && t + 1 < n_tiles) __builtin_amdgcn_s_waitcnt((2 * SPW) | (7 << 4));      // vmcnt(2 SPW) expcnt(7) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        ATTN_PROBE(0)
        if (t + NS - 1 < n_tiles) issue(t + NS - 1, st_nxt);
        const int st_use = st_cur;
        st_nxt = st_cur;                                  // the stage read now is the one refilled next
        st_cur = st_cur + 1 == NS ? 0 : st_cur + 1;
        if (!live) continue;            // this wave's 32 query rows are all padding: it only streams K/V for the others
        ATTN_PROBE(1)
        const char* sk = smem + st_use * STAGE;
        const char* sv = sk + TILE;

        {
            // ---- S^T[key][q] for 64 keys: all 8 K fragments in flight, then 8 MFMAs alternating the two 32-key accumulators ----
            i32x4_t kf[8];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    kf[ks * 2 + sub] = *reinterpret_cast<const i32x4_t*>(sk + tile_off(sub * 32 + ql, ks * 2 + hi));
            __builtin_amdgcn_sched_barrier(0);
            f32x16_t s[2];
            const float s_init = PRE ? -m_run : 0.f;             // PRE: m_run = the reference max the scores are taken against (0 before tile 0)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = s_init;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) s[sub] = mfma32<T>(kf[ks * 2 + sub], qf[ks], s[sub]);
            // ---- the V^T fragments of this tile: requested now, consumed after the softmax ----
            i32x4_t vf[8];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        vf[(sub * 2 + s2) * 2 + d] = *reinterpret_cast<const i32x4_t*>(sv + tile_off(d * 32 + ql, sub * 4 + s2 * 2 + hi));
            __builtin_amdgcn_sched_barrier(0);
            ATTN_PROBE(2)
            const bool need_mask = a.causal || (t * 64 + 64 > a.ntok);        // wave-uniform
            if (need_mask) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool ok = key < a.ntok && (!a.causal || key <= qrow);
                        s[sub][r] = ok ? s[sub][r] : -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            ATTN_PROBE(3)
            float lsum = 0.f;
            if constexpr (!PRE) {
                mx *= a.scale_log2e;
                const float m_new = fmaxf(m_run, mx);
                const bool grew = !__all(m_new == m_run);
                float alpha = 1.0f;
                if (grew) alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[sub][r], a.scale_log2e, -m_new));
                        s[sub][r] = p;
                        lsum += p;
                    }
                l_run = l_run * alpha + lsum;
                if (grew) {
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                }
            } else {
                // mx is this tile's max RELATIVE to m_run.  Refresh the reference when some row would exceed 2^THR (and on the first tile,
                // whose reference is the arbitrary 0): delta = how far the reference moves up, per query.
                constexpr float THR = 8.0f;
                if (t == 0 || __any(mx > THR)) {
                    const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);           // tile 0: O = l = 0, any finite factor will do
                    m_run += delta;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[sub][r] -= delta;
                    if (t > 0) {
#pragma unroll
                        for (int d = 0; d < 2; ++d)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
                        for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
                    }
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s[sub][r] = __builtin_amdgcn_exp2f(s[sub][r]);
                    }
            }
            i32x4_t pf[2][2];
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
            ATTN_PROBE(4)
            // ---- O^T[d][q] += V^T[d][keys] P^T[keys][q], the two d blocks alternating ----
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) o[d] = mfma32<T>(vf[(sub * 2 + s2) * 2 + d], pf[sub][s2], o[d]);
                    if constexpr (PRE) lacc = mfma32<T>(ones, pf[sub][s2], lacc);
                }
        }
        ATTN_PROBE(5)
    }
    ATTN_PROBE_DUMP

    // ---- normalise and store: lane holds O[q][d = dblk*32 + 8g + 4hi + 0..3] -------------------------
    const float l_tot = PRE ? lacc[0] : l_run + __shfl_xor(l_run, 32);      // the MFMA already summed both key halves
    const float inv = 1.0f / l_tot;
    if (a.lse2 && hi == 0 && qrow < a.ntok) a.lse2[(size_t)bh * a.npad + qrow] = m_run + __builtin_amdgcn_logf(l_tot);   // v_log_f32 = log2
    if (qrow < a.ntok) {
        const int b = bh / a.H, h = bh - b * a.H;
        uint16_t* orow = a.out + ((size_t)b * a.ntok + qrow) * (a.H * 64) + h * 64;
        // A row's 8-dim group g is split over the two half-waves (hi = 0: dims 8g .. 8g+3, hi = 1: 8g+4 .. 8g+7).  One v_permlane32_swap per
        // packed word on a PAIR of groups (g, g + 1) gives the lower lane all 16 bytes of group g and the upper lane all 16 bytes of group
        // g + 1: 4 stores of 16 bytes per lane instead of 8 of 8 (the store tail of a row-per-lane epilogue is bound by the number of store
        // instructions, cdna_hip_programming.md T21) -- same bytes at the same addresses.
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const uint32_t ax = pack2<T>(o[d][g * 4 + 0] * inv, o[d][g * 4 + 1] * inv), ay = pack2<T>(o[d][g * 4 + 2] * inv, o[d][g * 4 + 3] * inv);
                const uint32_t bx = pack2<T>(o[d][g * 4 + 4] * inv, o[d][g * 4 + 5] * inv), by = pack2<T>(o[d][g * 4 + 6] * inv, o[d][g * 4 + 7] * inv);
                const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);      // [0]: own g | lower's g+1, [1]: upper's g | own g+1
                const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                *reinterpret_cast<uint4*>(orow + d * 32 + (g + hi) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
    }
}

}  // namespace

int launch_attention(const void* q, const void* k, const void* vt, void* out, int B, int H, int ntok,
                     int npad, int dtype, int causal, float scale, hipStream_t stream) {
    return launch_attention_lse(q, k, vt, out, nullptr, B, H, ntok, npad, dtype, causal, scale, stream);
}

int launch_attention_lse(const void* q, const void* k, const void* vt, void* out, float* lse2, int B, int H, int ntok,
                         int npad, int dtype, int causal, float scale, hipStream_t stream) {
    return launch_attention_ex(q, k, vt, out, lse2, B, H, ntok, npad, dtype, causal, scale, 0, stream);
}

// prescaled != 0: q was written as T(q * scale * log2(e)) (one rounding, in the QKV GEMM's epilogue: GemmArgs::qkv_qscale) -- the
// inference engine's path; the matrix pipe then delivers the exp2 argument directly (kernel PRE = true).  Not for causal masks.
int launch_attention_ex(const void* q, const void* k, const void* vt, void* out, float* lse2, int B, int H, int ntok,
                        int npad, int dtype, int causal, float scale, int prescaled, hipStream_t stream) {
    if (npad % 128 != 0 || npad < ntok) return set_error(LSEG_ERR_INVALID, "attention: npad=%d must be a multiple of 128 and >= ntok=%d", npad, ntok);
    if (prescaled && causal) return set_error(LSEG_ERR_UNSUPPORTED, "attention: the pre-scaled form has no causal variant");
    AttnArgs a;
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.vt = (const uint16_t*)vt; a.out = (uint16_t*)out;
    a.lse2 = lse2;
    a.B = B; a.H = H; a.ntok = ntok; a.npad = npad; a.causal = causal;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.prescaled = prescaled;
    if (dtype != DT_BF16 && dtype != DT_F16) return set_error(LSEG_ERR_INVALID, "attention: dtype %d", dtype);
    int dev = 0;
    LSEG_HIP_TRY(hipGetDevice(&dev));
    const bool narrow = (long)((ntok + 127) / 128) * B * H < 2L * device_cu_count(dev) && !causal;      // 2-wave workgroups when the grid would not fill the chip
    const int nw = narrow ? 2 : 4;
    const size_t lds = (size_t)2 * 2 * 8192;
    dim3 grid(((ntok + 32 * nw - 1) / (32 * nw)) * B * H);
#define ATT(TT, NWV, PV) hipLaunchKernelGGL((lseg_attention_kernel<TT, NWV, 2, PV>), grid, dim3(64 * NWV), lds, stream, a)
#define ATT_T(NWV, PV) do { if (dtype == DT_BF16) ATT(BF16, NWV, PV); else ATT(F16, NWV, PV); } while (0)
    if (nw == 2) { if (prescaled) ATT_T(2, true); else ATT_T(2, false); }
    else { if (prescaled) ATT_T(4, true); else ATT_T(4, false); }
#undef ATT_T
#undef ATT
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace lseg

// corr.hip -- the pixel x text correlation of LSeg (modules/models/lseg_net.py:187-196) as ONE dedicated gfx950 kernel on the commuted
// schedule (engine.hip "commuted correlation"): label planes  R[b, k, p] = t_k . g_p  at the quarter resolution AND the 2x2-cell dot
// products of g ("gram", what the per-pixel norm of the up-sampled feature needs) from ONE pass over g.
//
// BASELINE north_star: "[B.H.W, C] x [K, C]^T label correlation ... LDS staging of ... text-embedding tiles".  Bound: HBM (per image at
// 122^2 padded pixels, C = 512, K = 150: g 15.2 MB read, R 8.9 MB written, gram 0.3 MB written; 2 x 150 x 512 flops per pixel = 95 flop/B).
// Round 4 ran it as two kernels that each read g -- the generic GEMM with the labels as rows (160x128 tiles, T re-streamed from L2 for
// every tile, K-loop / epilogue serialised in one workgroup per CU: 0.32 of the HBM rate) and pixel_gram_kernel (every g row read twice:
// 1.52x traffic, 0.36).
//
// Structure of the shipped kernel (v2; one workgroup = 4 waves = one per SIMD = one CU, persistent):
//   * T (the normalised text features, fp16 [K, 512]) is copied to LDS ONCE per workgroup and stays there for its life: K x 1040 bytes
//     (pitch 1024 + 16), 156 000 B of the CU's 160 KB at K = 150.  LDS holds nothing else.  Fragments of T run three k-steps ahead of their
//     MFMAs in a TD = 4 deep register ring (counted lgkmcnt waits).
//   * g is STREAMED ONCE from HBM straight into MFMA operand registers: a wave's tile = CORR_TR = 2 image rows x 16 columns of the padded map
//     (+ the row below and a 16th column as halo when the gram is wanted: 15 owned columns); a fragment = 16 pixels x 32 channels, lane
//     (pixel c, k-group kg) loads the 16 bytes g[pixel, 32 ks + 8 kg ..].  ONE wave per tile holds all NLB (<= 10) label blocks:
//     acc[NLB][2] accumulators, and a WHOLE tile of prefetch -- buf[8][2 (+1 halo)][..]: the 8 two-k-step groups of the NEXT tile are
//     requested group by group as the current tile's groups are consumed (counted vmcnt), ~416 of the wave's 512 registers.
//   * v_mfma_f32_16x16x32_f16 with the PIXELS as rows: a lane's 4 accumulator registers are 4 consecutive pixels of one label plane ->
//     16-byte plane stores, and the SAME instruction / operand roles / k order as the generic GEMM it replaces (csrc/gemm.hip,
//     MAP_LABELPLANES) -> the same bits.
//   * gram: the g fragments are A and B operand alike (same register layout), so the cell dot products are 4 more MFMAs per k-step on
//     operands that are already there: row r with itself (diagonal: g_q.g_q, first super-diagonal: g_q.g_(x+1)) and with row r+1 (diagonal:
//     g_q.g_(y+1,x), super: g_q.g_(y+1,x+1), sub: g_(x+1).g_(y+1,x)) -- the five records of elementwise.hip's norm_scale_plane_kernel.
//   * tiles are dealt so that neighbours in the image run at the same time on the same XCD (its L2 serves the halo re-reads).
// Measured and reverted (profiles/r05_corr_kernel.txt): v1 -- 4-row tiles, one two-k-step group of prefetch, all registers in accumulators:
// 357 us at B = 36 (latency-bound: 1024 waves x 10 KB in flight); TWO waves per tile for K > 80 (2 per SIMD, half the label blocks each on their
// own copy of the fragments): 40 % slower -- the pair loads the same bytes and the UNIQUE bytes in flight per CU halve.
#include "ops.h"
#include "gemm.h"
#include "../../include/lseg_hip.h"

#include <atomic>
#include <type_traits>
#include <utility>

namespace lseg {
namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int CORR_C = 512;                 // channels (out_c of clip_vitl16_384 / clip_vitb32_384); other widths take the generic GEMM
constexpr int CORR_KS = CORR_C / 32;        // k-steps of 32 channels
constexpr int CORR_NG = CORR_KS / 2;        // groups of two k-steps (= one 128-byte line of every pixel)
constexpr int CORR_PITCH = CORR_C * 2 + 16; // LDS row pitch of T in bytes
constexpr int CORR_TR = 2;                  // image rows of a wave's tile
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));    // a 16-byte plane store at a 4-byte aligned address (x0 = 1 + ...)

// NLB = label blocks of 16 (K <= 16 NLB)
template <int NLB, bool GRAM>
__global__ __launch_bounds__(256, 1) void corr_planes_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ T,
                                                            float* __restrict__ R, float* __restrict__ gram, int B, int K, int H, int W,
                                                            int tiles_y, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) char tlds[];
    constexpr int TR = CORR_TR;
    constexpr int ROWS = GRAM ? TR + 1 : TR;      // fragments per k-step: the tile rows (+ the halo row below)
    constexpr int XSTRIDE = GRAM ? 15 : 16;       // owned columns per tile
    const int HP = H + 2, WP = W + 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;

    // ---- T -> LDS, once ----------------------------------------------------------------------------------------------------------
    for (int i = tid; i < K * (CORR_C / 8); i += 256) {
        const int row = i / (CORR_C / 8), ch = i - row * (CORR_C / 8);
        *reinterpret_cast<i32x4_t*>(tlds + row * CORR_PITCH + ch * 16) = *reinterpret_cast<const i32x4_t*>(T + (size_t)row * CORR_C + ch * 8);
    }
    __syncthreads();
    int t_off[NLB];                               // this lane's T row (label 16 lb + c, clamped: rows >= K are computed and dropped)
#pragma unroll
    for (int lb = 0; lb < NLB; ++lb) {
        const int lab = lb * 16 + c;
        t_off[lb] = (lab < K ? lab : K - 1) * CORR_PITCH + kg * 16;
    }

    // ---- this wave's tiles: XCD-contiguous chunks, stride (waves of the XCD) inside ---------------------------------------------------
    const int ntiles = B * tiles_y * tiles_x;                                            // (launcher: < 2^31)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int chunk = (ntiles + 7) / 8;
    const int t_begin = xcd * chunk, t_end = t_begin + chunk < ntiles ? t_begin + chunk : ntiles;
    const int step = nslot * 4;
    int t = t_begin + slot * 4 + wave;

    // per-lane BYTE OFFSETS into g (32-bit: the launcher checks the map is < 4 GB; offsets from the kernel argument keep the loads in the
    // global address space -- loop-carried pointers came out as flat_load, which counts on vmcnt AND lgkmcnt and drained both) of the ROWS
    // fragment rows of tile `tt`; rows / columns clamped into the padded map.  The tile index is wave-uniform: one 32-bit decode per tile on
    // the scalar unit
    const char* gbase = reinterpret_cast<const char*>(g);
    auto frag_offs = [&](int tt, uint32_t (&out)[ROWS]) {
        const unsigned ut = (unsigned)__builtin_amdgcn_readfirstlane(tt);
        const unsigned q = ut / (unsigned)tiles_x, tx = ut - q * (unsigned)tiles_x;
        const unsigned b = q / (unsigned)tiles_y, ty = q - b * (unsigned)tiles_y;
        int x = 1 + XSTRIDE * (int)tx + c;
        x = x < WP - 1 ? x : WP - 1;
        const uint32_t col = ((uint32_t)b * (uint32_t)(HP * WP) + (uint32_t)x) * (CORR_C * 2) + (uint32_t)kg * 16u;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            int y = 1 + TR * (int)ty + r;
            y = y < HP - 1 ? y : HP - 1;
            out[r] = col + (uint32_t)(y * WP) * (CORR_C * 2);
        }
    };
    auto gload = [&](uint32_t off, int ks) { return *reinterpret_cast<const i32x4_t*>(gbase + off + ks * 64); };

    // The WHOLE tile lives in registers: buf[group][k-step of the group][fragment row].  Group j of the next tile is requested as soon as
    // group j of the current tile has been consumed, so a wave always has one full tile (16 x ROWS KB) in flight.
    i32x4_t buf[CORR_NG][2][ROWS];
    uint32_t fp[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) fp[r] = 0;
    if (t < t_end) {
        frag_offs(t, fp);
#pragma unroll
        for (int grp = 0; grp < CORR_NG; ++grp)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) buf[grp][kk][r] = gload(fp[r], grp * 2 + kk);
    }

    for (; t < t_end; t += step) {
        f32x4_t acc[NLB][TR];
        f32x4_t ga[GRAM ? TR : 1], gb[GRAM ? TR : 1];            // ga[r] = row r . row r, gb[r] = row r . row r + 1 (16 x 16 pixel blocks)
#pragma unroll
        for (int lb = 0; lb < NLB; ++lb)
#pragma unroll
            for (int r = 0; r < TR; ++r) acc[lb][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < (GRAM ? TR : 1); ++r) { ga[r] = f32x4_t{0.f, 0.f, 0.f, 0.f}; gb[r] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        const int tn = t + step;
        const bool more = tn < t_end;
        uint32_t np[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) np[r] = fp[r];
        if (more) frag_offs(tn, np);

        // T fragments run TD - 1 steps ahead of their MFMAs in a small register ring (a step = one label block of one k-step: one
        // ds_read_b128, TR MFMAs): with ONE wave per SIMD nothing else hides the LDS latency
        constexpr int TD = 4, NSTEP = CORR_KS * NLB;
        i32x4_t tf[TD];
        auto tread = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            return *reinterpret_cast<const i32x4_t*>(tlds + t_off[u % NLB] + (u / NLB) * 64);
        };
        static_for<0, TD - 1>([&](auto uc) { tf[decltype(uc)::value] = tread(uc); });
        static_for<0, CORR_NG>([&](auto gc) {
            constexpr int grp = decltype(gc)::value;
            static_for<0, 2>([&](auto kc) {
                constexpr int kk = decltype(kc)::value, ks = grp * 2 + kk;
                static_for<0, NLB>([&](auto lc) {
                    constexpr int lb = decltype(lc)::value, u = ks * NLB + lb;
                    if constexpr (u + TD - 1 < NSTEP) tf[(u + TD - 1) % TD] = tread(std::integral_constant<int, u + TD - 1>{});
#pragma unroll
                    for (int r = 0; r < TR; ++r) acc[lb][r] = mfma16<F16>(buf[grp][kk][r], tf[u % TD], acc[lb][r]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (GRAM) {
#pragma unroll
                    for (int r = 0; r < TR; ++r) {
                        ga[r] = mfma16<F16>(buf[grp][kk][r], buf[grp][kk][r], ga[r]);
                        gb[r] = mfma16<F16>(buf[grp][kk][r], buf[grp][kk][r + 1], gb[r]);
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            // this group's registers are free: the same group of the NEXT tile is requested (of this tile again after the wave's last one --
            // an unconditional load keeps the waits counted: behind a branch hipcc assumes nothing was issued and drains to vmcnt(0))
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) buf[grp][kk][r] = gload(np[r], grp * 2 + kk);
            __builtin_amdgcn_sched_barrier(0);
        });

        // ---- stores: label planes (interior pixels only: the x2 upsample reads nothing else), then the gram records ----------------
        const unsigned ut_ = (unsigned)__builtin_amdgcn_readfirstlane(t);
        const unsigned q_ = ut_ / (unsigned)tiles_x, tx = ut_ - q_ * (unsigned)tiles_x;
        const unsigned b = q_ / (unsigned)tiles_y, ty = q_ - b * (unsigned)tiles_y;
        const int y0 = 1 + TR * (int)ty, x0 = 1 + XSTRIDE * (int)tx;
        const size_t plane = (size_t)HP * WP;
        const int xs = x0 + 4 * kg;                                   // first of this lane's 4 pixels (D rows 4 kg .. 4 kg + 3)
#pragma unroll
        for (int lb = 0; lb < NLB; ++lb) {
            const int lab = lb * 16 + c;
            if (lab >= K) continue;
            float* pl = R + ((size_t)b * K + lab) * plane;
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const int y = y0 + r;
                if (y > H) continue;
                float* dst = pl + (size_t)y * WP + xs;
                const f32x4_t v = acc[lb][r];
                if (xs + 3 <= W) {
                    *reinterpret_cast<f32x4_u*>(dst) = v;             // global_store_dwordx4 needs dword alignment only
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (xs + e <= W) dst[e] = v[e];
                }
            }
        }
        if constexpr (GRAM) {
            // D[m][n] = g(row a, column m) . g(row b, column n); this lane: n = c, m = 4 kg + e
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const int y = y0 + r;
                if (y > H) continue;
                float* grow = gram + ((size_t)b * H + (y - 1)) * W * 5;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 4 * kg + e;
                    if (m <= 14 && x0 + m <= W) {                     // records of the owned pixel (y, x0 + m)
                        float* rec = grow + (size_t)(x0 + m - 1) * 5;
                        if (c == m) { rec[0] = ga[r][e]; rec[2] = gb[r][e]; }
                        if (c == m + 1) { rec[1] = ga[r][e]; rec[3] = gb[r][e]; }
                    }
                    if (c == m - 1 && c <= 14 && x0 + c <= W)          // g(y, x+1) . g(y+1, x) belongs to the pixel in column n = c
                        grow[(size_t)(x0 + c - 1) * 5 + 4] = gb[r][e];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) fp[r] = np[r];
    }
}

template <int NLB, bool GRAM>
int launch_corr(const void* g, const void* T, float* R, float* gram, int B, int K, int H, int W, hipStream_t st) {
    int dev = 0;
    LSEG_HIP_TRY(hipGetDevice(&dev));
    const int tiles_y = (H + CORR_TR - 1) / CORR_TR, tiles_x = GRAM ? (W + 14) / 15 : (W + 15) / 16;
    const long ntiles = (long)B * tiles_y * tiles_x;
    if (ntiles >= (1L << 31) || (size_t)B * (H + 2) * (W + 2) * CORR_C * 2 >= ((size_t)1 << 32))
        return set_error(LSEG_ERR_UNSUPPORTED, "corr_planes: B=%d %dx%d exceeds the kernel's 32-bit offsets", B, H, W);
    int grid = device_cu_count(dev) & ~7;
    if (grid < 8) grid = 8;
    const long want = ((ntiles + 3) / 4 + 7) / 8 * 8;                // no more workgroups than tiles / 4 (each copies T into its LDS)
    if (want < grid) grid = (int)want;
    const size_t lds = (size_t)K * CORR_PITCH;
    auto kern = corr_planes_kernel<NLB, GRAM>;
    static std::atomic<unsigned long long> attr_done{0};             // per-device opt-in to > 64 KB of dynamic LDS (cf. gemm.hip)
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const uint16_t*)g, (const uint16_t*)T, R, gram, B, K, H, W, tiles_y, tiles_x);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

bool corr_planes_supported(int K, int C) { return C == CORR_C && K >= 1 && (size_t)K * CORR_PITCH <= 160u * 1024u; }

// g: padded NHWC fp16 [B, H+2, W+2, 512]; T: fp16 [K, 512]; R: fp32 [B, K, (H+2)(W+2)] (interior written); gram: fp32 [B, H, W, 5] or NULL
int launch_corr_planes(const void* g, const void* T, float* R, float* gram, int B, int K, int H, int W, int C, hipStream_t st) {
    if (!corr_planes_supported(K, C)) return set_error(LSEG_ERR_UNSUPPORTED, "corr_planes: K=%d C=%d (C must be 512, K x 1040 B must fit the LDS)", K, C);
    if (B < 1 || H < 1 || W < 1) return set_error(LSEG_ERR_INVALID, "corr_planes: B=%d H=%d W=%d", B, H, W);
#define CORR_GO(NLB) (gram ? launch_corr<NLB, true>(g, T, R, gram, B, K, H, W, st) : launch_corr<NLB, false>(g, T, R, nullptr, B, K, H, W, st))
    if (K <= 32) return CORR_GO(2);
    if (K <= 80) return CORR_GO(5);
    return CORR_GO(10);
#undef CORR_GO
}

}  // namespace lseg

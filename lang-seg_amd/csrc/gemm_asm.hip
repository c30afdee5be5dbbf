// gemm_asm.hip -- hand-scheduled gfx950 GEMM for the ViT block's residual Linears (attn.proj, mlp.fc2: timm Block via lseg_vit.py:196-197):
//     C[M, N] (fp32, in place) += A[M, K] W[N, K]^T + bias
// The device code is ONE asm statement per operand type, written by gemm_asm_gen.py (schedule, register map and the reasons are in its
// header); this file is the kernel shell around it, the host-side tile list and the launcher.
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "gemm.h"
#include "../../include/lseg_hip.h"
#include "build/gemm_asm_body.inc"

namespace lseg {
namespace {

struct AsmGemmArgs {              // read by the asm body with s_load_dwordx8 x 2: offsets are part of the contract with gemm_asm_gen.py
    const void* A;                // 0
    const void* W;                // 8
    void* C;                      // 16   fp32 [rows_alloc, N], read (residual) and written
    const float* bias;            // 24
    const uint32_t* tiles;        // 32   [grid][LSEG_GEMM_ASM_TILE_SLOTS] entries mb << 16 | nb, 0xffffffff = none
    void* scratch;                // 40   a throw-away tile (the pipeline's "tile -1" stores and its "tile n" residual loads)
    int nk;                       // 48   K / 64
    int lda_bytes;                // 52   row pitch of A and of W
    int ldc_bytes;                // 56
    int flags;                    // 60
};
static_assert(sizeof(AsmGemmArgs) == 64, "kernarg layout");

// 4 waves (one per SIMD, the whole 512-entry register file each) or 8 (two per SIMD, 256 registers each): gemm_asm_gen.py GEMM_ASM_WAVES
__global__ __launch_bounds__(LSEG_GEMM_ASM_THREADS) __attribute__((amdgpu_waves_per_eu(LSEG_GEMM_ASM_WAVES_PER_EU, LSEG_GEMM_ASM_WAVES_PER_EU))) void lseg_gemm_res32_asm_f16(const AsmGemmArgs a) {
    asm volatile(LSEG_GEMM_ASM_BODY_F16
                 :
                 : "s"((uint64_t)__builtin_amdgcn_kernarg_segment_ptr()), "s"(blockIdx.x), "v"(threadIdx.x)
                 : LSEG_GEMM_ASM_CLOBBERS);
}
__global__ __launch_bounds__(LSEG_GEMM_ASM_THREADS) __attribute__((amdgpu_waves_per_eu(LSEG_GEMM_ASM_WAVES_PER_EU, LSEG_GEMM_ASM_WAVES_PER_EU))) void lseg_gemm_res32_asm_bf16(const AsmGemmArgs a) {
    asm volatile(LSEG_GEMM_ASM_BODY_BF16
                 :
                 : "s"((uint64_t)__builtin_amdgcn_kernarg_segment_ptr()), "s"(blockIdx.x), "v"(threadIdx.x)
                 : LSEG_GEMM_ASM_CLOBBERS);
}

// ---- tile lists: XCD-aware grouped order, one list per workgroup (same rasterisation as gemm.hip's persistent kernels: XCD x owns a
// contiguous range of the grouped order, its workgroups walk it with stride grid / 8) ----------------------------------------------------
struct TileList { uint32_t* dev = nullptr; int grid = 0; };
struct DevState {
    std::mutex mu;
    std::map<std::tuple<int, int, int>, TileList> lists;      // (tiles_m, tiles_n, grid cap)
    void* scratch = nullptr; size_t scratch_bytes = 0;
};
DevState g_dev[64];

int build_list(int tiles_m, int tiles_n, int max_grid, int cus, TileList& out) {
    const int total = tiles_m * tiles_n;
    int grid = ((total + 7) / 8) * 8;
    if (grid > cus) grid = cus & ~7;
    if (max_grid >= 8 && grid > max_grid) grid = max_grid & ~7;
    if (grid < 8) grid = 8;
    const int wpx = grid >> 3, GROUP_M = 8;
    std::vector<uint32_t> h((size_t)grid * LSEG_GEMM_ASM_TILE_SLOTS, 0xffffffffu);
    for (int b = 0; b < grid; ++b) {
        const int xcd = b & 7, idx = b >> 3;
        const int q = total >> 3, r = total & 7;
        const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        int n = 0;
        for (int t = xs + idx; t < xs + cnt; t += wpx) {
            if (n == LSEG_GEMM_ASM_TILE_SLOTS) return -1;
            const int per_group = GROUP_M * tiles_n, gid = t / per_group, first_m = gid * GROUP_M;
            const int gsz = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
            const int rr = t - gid * per_group;
            const int mb = first_m + rr % gsz, nb = rr / gsz;
            h[(size_t)b * LSEG_GEMM_ASM_TILE_SLOTS + n++] = ((uint32_t)mb << 16) | (uint32_t)nb;
        }
    }
    uint32_t* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(uint32_t)) != hipSuccess) return -2;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return -2; }
    out.dev = d; out.grid = grid;
    return 0;
}

}  // namespace

// Opt-in (LSEG_GEMM_ASM=1): measured on MI355X at B = 36 (profiles/r06_gemm_asm.txt) the 256 x 128 double-accumulator schedule hides the
// residual epilogue completely but pays 1.5x the operand bytes per flop of the 256 x 256 tiles -- attn.proj 103 vs 98 us, mlp.fc2 287 vs 252 us
// against the generic kernel family -- so the engine keeps the generic kernels unless asked.
bool gemm_res32_asm_enabled() {
    static const int on = getenv("LSEG_GEMM_ASM") ? atoi(getenv("LSEG_GEMM_ASM")) == 1 : 0;
    return on != 0;
}

// Can launch_gemm_res32_asm run this problem?  (The caller falls back to the generic kernel family otherwise.)
bool gemm_res32_asm_eligible(const GemmArgs& g, int ab_dtype) {
    if (ab_dtype != DT_F16 && ab_dtype != DT_BF16) return false;
    if (g.conv || g.kmajor || g.split || g.nsplit > 1 || g.relu_in || g.round_mid || g.act != ACT_NONE || g.map_mode != MAP_LINEAR) return false;
    if (g.res_mode != RES_DEST || g.res != g.C || g.res2 || g.res_dtype != DT_F32 || g.out_dtype != DT_F32 || !g.bias || g.bias_mod) return false;
    if ((g.N % 128) || (g.K % 64) || (g.K >> 6) < LSEG_GEMM_ASM_MIN_KSTEPS || g.lda != g.K || g.ldw != g.K || g.ldc != g.N) return false;
    const int rows = (g.M + 255) / 256 * 256;
    if (g.rows_alloc < rows) return false;                         // A and C must be readable / writable up to the tile boundary
    if ((double)rows * g.K * 2 >= 4.0e9 || (double)g.N * g.K * 2 >= 4.0e9 || (double)256 * g.N * 4 >= 2.0e9) return false;    // 32-bit lane offsets
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.bias)) & 15) return false;
    if (g.dbg & 7) return false;
    return true;
}

int launch_gemm_res32_asm(const GemmArgs& g, int ab_dtype, hipStream_t stream) {
    int dev = 0;
    LSEG_HIP_TRY(hipGetDevice(&dev));
    DevState& S = g_dev[dev & 63];
    const int tiles_m = (g.M + 255) / 256, tiles_n = g.N / 128;
    TileList tl;
    {
        std::lock_guard<std::mutex> lk(S.mu);
        const auto key = std::make_tuple(tiles_m, tiles_n, g.max_grid);
        auto it = S.lists.find(key);
        if (it == S.lists.end()) {
            TileList n;
            const int rc = build_list(tiles_m, tiles_n, g.max_grid, device_cu_count(dev), n);
            if (rc == -1) return set_error(LSEG_ERR_UNSUPPORTED, "gemm (asm): more than %d tiles per workgroup", LSEG_GEMM_ASM_TILE_SLOTS);
            if (rc) return set_error(LSEG_ERR_HIP, "gemm (asm): tile list allocation failed");
            it = S.lists.emplace(key, n).first;
        }
        tl = it->second;
        const size_t need = (size_t)256 * g.N * 4 + 4096;
        if (S.scratch_bytes < need) {
            // (an older, smaller scratch tile is left allocated: launches already enqueued may still use it)
            void* p = nullptr;
            LSEG_HIP_TRY(hipMalloc(&p, need));
            LSEG_HIP_TRY(hipMemset(p, 0, need));
            S.scratch = p; S.scratch_bytes = need;
        }
    }
    AsmGemmArgs a;
    a.A = g.A; a.W = g.W; a.C = g.C; a.bias = g.bias; a.tiles = tl.dev; a.scratch = S.scratch;
    a.nk = g.K >> 6; a.lda_bytes = g.K * 2; a.ldc_bytes = g.N * 4; a.flags = 0;
    auto kern = ab_dtype == DT_F16 ? lseg_gemm_res32_asm_f16 : lseg_gemm_res32_asm_bf16;
    static std::atomic<unsigned long long> attr_done[2];
    const unsigned long long bit = 1ull << (dev & 63);
    std::atomic<unsigned long long>& done = attr_done[ab_dtype == DT_F16 ? 0 : 1];
    if (!(done.load(std::memory_order_acquire) & bit)) {
        LSEG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LSEG_GEMM_ASM_LDS));
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(tl.grid), dim3(LSEG_GEMM_ASM_THREADS), LSEG_GEMM_ASM_LDS, stream, a);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace lseg

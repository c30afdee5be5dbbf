// gemm.h -- argument block of the MFMA GEMM / implicit-GEMM-conv kernel family.
#pragma once
#include "common.h"

namespace lseg {

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_QUICKGELU = 2, ACT_RELU = 3 };

// Where output element (m, n) goes.
enum MapMode {
    MAP_LINEAR = 0,   // C[m*ldc + n]
    MAP_PERIODIC = 1, // row' = (m / p_div) * p_mul + (m % p_div) + p_off ; C[row'*ldc + n]
    MAP_PADDED = 2,   // m=(b,y,x) on (Ho,Wo): NHWC with a 1-pixel zero border, C channels = ldc
    MAP_PIXSHUF = 3,  // ConvTranspose k=s: m=(b,y,x) on (Ho,Wo); n=(i*s+j)*ps_C+co -> padded NHWC
                      // [b, y*s+i+1, x*s+j+1, co] of a (Ho*s+2, Wo*s+2) map
    MAP_QKV = 4,      // n=(which, head, d), m=(b,t): q,k -> [b,head,t,d] ; v -> [b,head,d,t]
    MAP_NCHW = 5,     // m=(b,p), P=p_div pixels: C[(b*N + n)*P + p]  (label-major logits planes)
    MAP_LABELPLANES = 7, // correlation, labels as GEMM rows: m = label, n = (b, p) with P = p_div pixels per image:
                      // C[(b*M + m)*P + p] -- the [B, K, h*w] logits planes, 4 consecutive pixels per lane (16-byte stores)
    MAP_ROWNORM = 6,  // fused head (N == 512): C[m*ldc+n] = fp16(rn_scale * fp16(v / ||v_row||_2)), v = acc + bias
};
enum ResMode { RES_NONE = 0, RES_DEST = 1, RES_PERIODIC = 2 };

struct GemmArgs {
    // operands: A activations, W weights [N, K] row-major (PyTorch Linear layout)
    const uint16_t* A;
    const uint16_t* W;
    int M, N, K;
    int lda, ldw;
    // implicit-GEMM conv (A is padded NHWC [B, Hp, Wp, Cin]); K = taps * Cin
    int conv;               // 0 plain, 1 conv
    int cin, hp, wp;        // padded input geometry
    int ho, wo;             // output spatial size (also used by MAP_PADDED / MAP_PIXSHUF)
    int stride;             // conv stride (1|2)
    int relu_in;            // apply ReLU to A fragments (bf16 only)
    // epilogue
    const float* bias;      // fp32 [N] (or [bias_mod]) or null
    int bias_mod;           // 0: bias[n]; else bias[n % bias_mod]
    int round_mid;          // round (acc + bias) to the operand type before act/residual
                            // (reproduces CLIP's fp16 Linear -> fp16 add, [3P] clip/model.py)
    int act;
    int res_mode;
    const void* res;  int res_dtype; int ldr;   // residual tensor
    const void* res2;                            // optional second residual (RES_DEST, same dtype/geometry)
    void* C;  int out_dtype; int ldc;
    void* C_relu;               // MAP_PADDED on the specialised epilogue only (gemm_epilogue_is_pad16): a second copy ReLU(C), same geometry --
                                // the DPT residual units read ReLU(x) as conv input and x as skip (lseg_blocks.py:270-288): clamping the
                                // A fragments in the K-loop instead costs the 3x3 convs a third of their MFMA rate
    // training-step fusions of the MLP's GELU (plain MAP_LINEAR 16-bit GEMMs on the specialised epilogue only; gemm_fuses_gelu tells):
    void* C_pre;                // with act = ACT_GELU: also store the 16-bit pre-activation z = T(acc + bias) here (same geometry as C); C gets
                                // T(gelu(z)) of the ROUNDED z, so the pair equals a Linear followed by a GELU pass over its stored output
    const void* dgelu_pre;      // C = T((acc + bias) * gelu'(z)), z read from here (16-bit, same geometry as C): dX of fc2 through the GELU
    int map_mode;
    int p_div, p_mul, p_off;    // MAP_PERIODIC / RES_PERIODIC / MAP_NCHW(P)
    int ps_s, ps_C;             // MAP_PIXSHUF
    // MAP_QKV
    void* Ck; void* Cv; int qkv_dim, qkv_ntok, qkv_npad, qkv_heads;
    float qkv_qscale;           // != 0: the q third is written as T((acc + bias) * qkv_qscale) -- the attention's softmax scale * log2(e) folded
                                // into the epilogue's single rounding (specialised QKV epilogue only: gemm_qkv_scales_q tells)
    float rn_scale;             // MAP_ROWNORM: logit scale
    int tag;                    // 0 generic, 1 = the profiled dominant instance (distinct symbol)
    int dbg;                    // tools only: bits 0-1 ablation (1 = skip epilogue stores, 2 = skip the epilogue), 4 = phase timing, 8 = no stagger
    int group_m;                // row-blocks per rasterisation group (0 -> 8)
    // ---- split-precision ("strict") mode: every 16-bit operand is a (hi, lo) fp16 pair, x = hi + lo with lo = fp16(x - hi),
    // kept as two planes `plane` elements apart.  The K-loop runs three segments into the same fp32 accumulators:
    //   A_hi.W_hi + A_lo.W_hi + A_hi.W_lo   (the lo.lo term is below fp32 round-off) -- ~21 mantissa bits instead of 11.
    // 16-bit outputs are written as (hi, lo) planes as well; 16-bit residuals are read as hi + lo.
    int split;                  // 0 off, 1 on
    size_t a_plane, w_plane;    // element offset of the lo plane of A / W
    size_t c_plane, ck_plane, cv_plane, res_plane;   // lo planes of C / Ck / Cv / res (and res2)
    // ---- split-K: the K-steps are cut into `nsplit` ranges of `split_steps` (64-wide) steps; range s of every output tile is its own
    // work item and writes a PARTIAL result at C + s * c_split_stride (the caller sums the partials).  For contractions with a small
    // output and a huge K (weight gradients: [256 x 2304] over K = 119,072): fills the chip with 128x128 tiles instead of 64x64 ones,
    // whose operand bytes per flop are twice as high on the per-CU L2->LDS path.  Needs bias == NULL, no residual, MAP_LINEAR.
    int nsplit, split_steps;
    size_t c_split_stride;
    // ---- K-major operands (weight gradients, dW = dY^T X): A is [K, lda] with the M output rows contiguous, W is [K, ldw] with the N
    // output columns contiguous -- the operands as they sit in memory, no transposed copies.  Rows k >= k_valid read as zero (the K-steps
    // cover K = k_valid rounded up to 64).  128x128 tiles, fp32 split-K slabs (nsplit >= 1) as output; needs N % 128 == 0.
    int kmajor, k_valid;
    // ... of a 3x3 conv (kconv_cin > 0): W = the padded NHWC input [k_valid, kconv_cin], output column n = (tap, ci) reads W row
    // k + (tap / 3 - 1) * kconv_wp + (tap % 3 - 1) (rows outside [0, k_valid) are zero); relu_in clamps the W operand (the RCU convs read
    // ReLU(x)).  Needs kconv_cin % 128 == 0 (a 128-wide column tile lies inside one tap).
    int kconv_cin, kconv_wp;
    int max_grid;               // tools: cap of the persistent grid (workgroups, multiple of 8; 0 = the whole chip)
    int rows_alloc;             // rows of A and of C / res that are ALLOCATED (>= M; 0 = exactly M).  A caller whose buffers reach the next
                                // multiple of 256 rows lets the residual GEMMs run on the hand-scheduled kernel (gemm_asm.hip), which computes
                                // whole 256-row tiles and never masks: rows >= M are garbage in, garbage out, nobody reads them
    int tile_hint;              // 0: the launcher's cost model picks the tile; 2 = 128x128, 6 = 256x256 (callers that plan tile and split-K together)
};

void gemm_args_init(GemmArgs& g);
int device_cu_count(int dev);
// ab_dtype: DT_BF16 | DT_F16.  Returns 0 or a negative lseg_status.
int launch_gemm(const GemmArgs& g, int ab_dtype, hipStream_t stream);
// true when launch_gemm would run this problem on the specialised padded-NHWC epilogue (the one that honours C_relu)
bool gemm_epilogue_is_pad16(const GemmArgs& g, int ab_dtype);
// true when launch_gemm would honour g.qkv_qscale (it refuses a non-zero scale otherwise)
bool gemm_qkv_scales_q(const GemmArgs& g, int ab_dtype);
// true when launch_gemm would honour g.C_pre / g.dgelu_pre (it refuses them otherwise)
bool gemm_fuses_gelu(const GemmArgs& g, int ab_dtype);
// gemm_asm.hip: C (fp32, in place) += A W^T + bias on the hand-scheduled 256 x 128 kernel; launch_gemm takes it when eligible AND enabled (LSEG_GEMM_ASM=1)
bool gemm_res32_asm_eligible(const GemmArgs& g, int ab_dtype);
bool gemm_res32_asm_enabled();          // LSEG_GEMM_ASM=1
int launch_gemm_res32_asm(const GemmArgs& g, int ab_dtype, hipStream_t stream);

}  // namespace lseg

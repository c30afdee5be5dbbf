// ops.h -- host-side launchers of every kernel in the engine (all stream-ordered, return 0 or <0).
#pragma once
#include "common.h"
#include "gemm.h"

namespace lseg {

int launch_attention(const void* q, const void* k, const void* vt, void* out, int B, int H, int ntok,
                     int npad, int dtype, int causal, float scale, hipStream_t stream);

int launch_layernorm(const void* in, int in_dtype, const float* gamma, const float* beta, void* out, int out_dtype,
                     int M, int D, float eps, hipStream_t st);
// x += bias + sum of `nsplit` split-K partial slabs (`stride` floats apart), then LayerNorm of the updated rows (gamma NULL: update only)
int launch_layernorm_reduce(float* x, const float* part, int nsplit, size_t stride, const float* bias, const float* gamma, const float* beta,
                            void* out, int out_dtype, int M, int D, float eps, hipStream_t st);
// epilogue of a split-K 3x3 conv: sum of `ns` fp32 slabs [B*Ho*Wo, C] + bias, ReLU, skip inputs -> padded NHWC map (+ its ReLU copy)
int launch_conv_reduce_pad(const float* part, int ns, size_t stride, const float* bias, const void* res, const void* res2, void* out, void* out_relu,
                           int B, int Ho, int Wo, int C, int relu, int dtype, hipStream_t st);
int launch_im2col_patch(const float* x, void* A, int B, int H, int W, int P, int dtype, hipStream_t st);
int launch_pos_resize(const float* pos, float* out, int g_old, int gh, int gw, int D, hipStream_t st);
int launch_cls_rows(const float* cls, const float* pos, float* x, int B, int ntok, int D, hipStream_t st);
int launch_readout_cat(const float* x, void* A, int B, int ntok, int D, int dtype, hipStream_t st);
int launch_upsample2x_nhwc(const void* in, void* out, int B, int H, int W, int C, int dtype, hipStream_t st);
int launch_upsample2x_planes(const float* in, float* out, int P, int H, int W, hipStream_t st);
int launch_combine_1x1(const float* wh, const float* bh, const float* wo, const float* bo, float* wc, float* bc, int Co, int Cm, int Ci,
                       hipStream_t st);
int launch_pixel_gram(const void* g16, float* gram, int B, int H, int W, int C, hipStream_t st);
int launch_norm_scale_plane(const float* gram, float* scale, int B, int H, int W, float s, hipStream_t st, unsigned* flag = nullptr);
int launch_upsample2x_planes_scaled(const float* in_padded, const float* scale, float* out, int P, int K, int H, int W, hipStream_t st);
int launch_upsample4x_planes_scaled(const float* in_padded, const float* scale, float* out, int P, int K, int H, int W, hipStream_t st);
int launch_upsample_norm_f16(const float* g, void* a, int B, int H, int W, int C, float scale, hipStream_t st);
int launch_l2norm_scale_f16(const float* f, void* a, int M, int C, float scale, hipStream_t st);
int launch_text_embed(const int64_t* tok, const float* emb, const float* pos, void* x, int rows, int L, int ctx, int W, hipStream_t st);
int launch_text_pool(const void* x, const int* eot, void* pooled, int K, int L, int W, hipStream_t st);
int launch_text_l2norm(const void* t, void* out, int K, int C, hipStream_t st);
int launch_convert(const void* in, int in_dtype, void* out, int out_dtype, size_t n, hipStream_t st);
int launch_range16(const void* p, size_t n, int dtype, unsigned long long* out4, hipStream_t st);
int launch_convert2d(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, int ld, hipStream_t st);
int launch_transpose_convert(const void* in, int in_dtype, void* out, int out_dtype, int R, int C, hipStream_t st);
int launch_pack_conv3x3(const float* w, const float* bn_w, const float* bn_b, const float* bn_m, const float* bn_v,
                        float bn_eps, const float* conv_bias, void* wp, float* bias_out, int Co, int Ci, int Cip, int dtype,
                        hipStream_t st);      // dtype DT_F32: wp is a float buffer (split-precision packs go through fp32)
int launch_pack_convT(const float* w, void* wp, int Ci, int Co, int Cp, int s, int dtype, hipStream_t st);
int launch_nhwc_to_nchw_f32(const void* in, float* out, int B, int H, int W, int C, int Cs, int pad, int dtype, hipStream_t st, size_t lo_plane = 0);
int launch_rows_to_nchw_f32(const float* in, float* out, int B, int HW, int C, hipStream_t st);
int launch_head_block(const float* in, float* out, const float* w9, const float* bias, int B, int K, int H, int W,
                      int bottleneck, int act, int apply_act, hipStream_t st);
int launch_argmax_planes(const float* in, uint8_t* out, int B, int K, int HW, hipStream_t st);
int launch_layernorm_backward(const void* dy, int dy_dtype, const float* x, const float* gamma, float* dx, float* dgamma,
                              float* dbeta, int M, int D, float eps, int accumulate, hipStream_t st, int accumulate_params = 0,
                              float* partial_ws = nullptr,       // partial_ws: [LN_BWD_PARTIAL_BLOCKS, 2D] floats -> no atomics
                              void* dx16 = nullptr);             // optional 16-bit copy (dy's type) of the updated dx
constexpr int LN_BWD_PARTIAL_BLOCKS = 1024;
int launch_transpose16(const void* in, void* out, int R, int C, int ldi, int ldo, hipStream_t st, int shift = 0, int relu = 0);
int launch_bn_train_forward(const void* x, void* y, float* stats, const float* gamma, const float* beta, int B, int H, int W, int C,
                            float eps, int dtype, hipStream_t st);
int launch_bn_train_backward(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, float* bstats,
                             int B, int H, int W, int C, float eps, int dtype, hipStream_t st);
int launch_relu_backward(const void* dy, const void* x, void* dx, size_t n, hipStream_t st);
int launch_upsample2x_planes_backward_rows(const float* dout, void* rows, int B, int K, int H, int W, int ldk, int dtype, hipStream_t st);
int launch_l2norm_scale_backward(const void* da, int da_dtype, const float* x, void* dx, int dx_dtype, int M, int C, float scale, hipStream_t st);
int launch_gelu_backward(const void* dy, const void* pre, void* dx, size_t n, int dtype, hipStream_t st, int quick = 0);
int launch_upsample2x_nhwc_backward(const void* dout, void* din, int B, int H, int W, int C, int dtype, hipStream_t st);
int launch_softmax_ce_backward(const float* scores, const int64_t* target, float* dz, int B, int K, int HW, int ignore_index,
                               const double* nll, hipStream_t st);
int launch_conv_dgrad_pack(const void* wp, void* wd, int Co, int Ci, hipStream_t st);
// det_ws (optional, det_cap floats): deterministic reductions -- partial rows + a fixed-order sum instead of fp32 atomics
int launch_colsum16(const void* in, int dtype, float* out, int R, int C, int ld, hipStream_t st, int accumulate = 0,
                    float* det_ws = nullptr, size_t det_cap = 0);
int launch_relu_backward_add(const void* dy, const void* x, const void* add, void* dx, size_t n, int dtype, hipStream_t st);
int launch_seg_stats_ex(const float* scores, const int64_t* target, int B, int K, int HW, int ignore_index, unsigned long long* counts,
                        double* nll, uint8_t* argmax_out, int up, int h, int w, hipStream_t st, float* lse_out = nullptr);
int launch_upsample_ce_backward_rows(const float* low, const int64_t* target, const float* lse, const double* nll, void* rows, int B, int K,
                                     int H, int W, int ldk, int ignore_index, int dtype, hipStream_t st, const float* gscale = nullptr);
int launch_seg_stats(const float* scores, const int64_t* target, int B, int K, int HW, int ignore_index,
                     unsigned long long* counts, double* nll, hipStream_t st);

// ---- split-precision ("strict") mode (strict.hip): 16-bit tensors as (hi, lo) fp16 planes `plane` elements apart --------------
int launch_convert_split(const void* in, int in_dtype, void* out, size_t n, size_t plane, hipStream_t st);
int launch_ln_split(const float* x, const float* gamma, const float* beta, void* out, size_t plane, int M, int D, float eps, hipStream_t st);
int launch_im2col_split(const float* x, void* A, size_t plane, int B, int H, int W, int P, hipStream_t st);
int launch_readout_cat_split(const float* x, void* A, size_t plane, int B, int ntok, int D, hipStream_t st);
int launch_upsample2x_nhwc_split(const void* in, size_t in_plane, void* out, size_t out_plane, int B, int H, int W, int C, hipStream_t st);
int launch_relu_split(const void* in, size_t in_plane, void* out, size_t out_plane, size_t n, hipStream_t st);
int launch_attention_strict(const void* q, const void* k, const void* vt, void* out, size_t qk_plane, size_t vt_plane, size_t out_plane,
                            int B, int H, int ntok, int npad, float scale, hipStream_t st);

size_t attention_backward_ws_bytes(int B, int H, int npad);
int launch_attention_backward_qkv(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse2, void* dqkv,
                                  void* ws, int B, int H, int ntok, int npad, int dtype, float scale, hipStream_t stream);

// dW [rows_out, cols_out] (+)= dY[:, :rows_out]^T X[:, :cols_out], contraction over the M rows of dY [M, ldy] / X [M, ldx] (16-bit), through the
// K-major operand path of the GEMM (no transposed copies); `ws` = fp32 scratch for the split-K slabs (ws_floats >= rows_out * cols_out).
// Returns LSEG_ERR_UNSUPPORTED (and launches nothing) when cols_out % 128 != 0: callers keep the transposing path for those.
bool wgrad_kmajor_ok(int rows_out, int cols_out, size_t ws_floats);
int launch_wgrad_kmajor(const void* dy, int ldy, const void* x, int ldx, int M, int rows_out, int cols_out, float* dw, int accumulate,
                        float* ws, size_t ws_floats, int ab_dtype, hipStream_t st);

// 3x3 conv weight gradient in the padded-NHWC layout through the same path: slabs [ns][Cout][9 * Cin] (tap-major) in `ws`; *ns_out = slab count
// (the caller sums / re-layouts them: launch_conv_wgrad_unpack / launch_sum_partials).  Needs Cin % 128 == 0.
int launch_conv_wgrad_kmajor(const void* dy_pad, const void* x_pad, int relu_x, int B, int H, int W, int Cin, int Cout, float* ws, size_t ws_floats,
                             int ab_dtype, int* ns_out, hipStream_t st);

// ---- engine-level training step (train.hip) --------------------------------------------------------------------------------
int launch_attention_lse(const void* q, const void* k, const void* vt, void* out, float* lse2, int B, int H, int ntok, int npad, int dtype,
                         int causal, float scale, hipStream_t stream);
int launch_attention_ex(const void* q, const void* k, const void* vt, void* out, float* lse2, int B, int H, int ntok, int npad, int dtype,
                        int causal, float scale, int prescaled, hipStream_t stream);
int launch_bn_stats(const void* x, float* stats, int B, int H, int W, int C, int dtype, hipStream_t st, int pre_zeroed = 0,
                    float* det_ws = nullptr, size_t det_cap = 0);
int launch_bn_apply(const void* x, void* y, const float* stats, const float* gamma, const float* beta, const void* res1, const void* res2,
                    int B, int H, int W, int C, float eps, double count, int dtype, hipStream_t st,
                    void* y_relu = nullptr);      // y_relu: ReLU(y), same geometry (y may then be NULL)
int launch_bn_bwd_stats(const void* dy, const void* x, const float* stats, float* bstats, int B, int H, int W, int C, float eps,
                        double count, int dtype, hipStream_t st, float* det_ws = nullptr, size_t det_cap = 0);
int launch_bn_bwd_apply(const void* dy, const void* x, const float* stats, const float* bstats, const float* gamma, void* dx,
                        int B, int H, int W, int C, float eps, double count, int dtype, hipStream_t st);
// corr.hip: dedicated pixel x text correlation on the commuted schedule (label planes + cell dot products of g from one pass)
bool corr_planes_supported(int K, int C);
int launch_corr_planes(const void* g16pad, const void* T16, float* R, float* gram, int B, int K, int H, int W, int C, hipStream_t st);
int launch_bn_running_update(const float* stats, float* rmean, float* rvar, int C, double count, float momentum, hipStream_t st);
int launch_gelu_forward(const void* pre, void* out, size_t n, int dtype, hipStream_t st);
int launch_unpad_rows(const void* in, void* out, int B, int H, int W, int C, hipStream_t st);
int launch_dilate2(const void* dy, void* dyd, int B, int Ho, int Wo, int H, int W, int C, hipStream_t st);
int launch_unpixshuf(const void* dl, void* dg, int B, int gh, int gw, int s, int C, hipStream_t st);
int launch_readout_cat_bwd(const void* dcat, float* gx, int B, int ntok, int D, int dtype, hipStream_t st);
int launch_embed_bwd(const float* gx, void* dtok, float* dpos, int B, int ntok, int D, int dtype, hipStream_t st);
int launch_pos_resize_bwd(const float* dpos, float* dposemb, float* dcls, int g_old, int gh, int gw, int D, hipStream_t st);
int launch_conv_wgrad_unpack(const float* dw, float* dst, int Co, int Ci, int Cip, int accumulate, hipStream_t st, int nsplit = 1, size_t split_stride = 0);
int launch_sum_partials(const float* part, float* dst, int nsplit, size_t n, size_t stride, int accumulate, hipStream_t st);
int launch_convT_wgrad_unpack(const float* dw, float* dst, int C, int Cp, int s, int accumulate, hipStream_t st);
int launch_fold_rows(const float* src, float* dst, int R, int C, int ld, int accumulate, hipStream_t st);
int launch_sgd(float* w, const float* g, float* m, void* w16, size_t n, float lr, float mu, float wd, int first, int dtype, hipStream_t st);
// table-driven launches of the optimizer step (tables live in device memory, sorted by blk0)
struct SgdSeg {
    float* w; const float* g; float* m;      // fp32 master, gradient, momentum
    uint16_t* w16; float* w32;               // the engine's same-layout copies of the parameter (either may be NULL)
    unsigned long long n;
    unsigned blk0;                           // first block of this parameter (4096 elements per block)
    int scratch;                             // learning-rate group: 0 = pretrained.*, 1 = scratch.*  (lsegmentation_module.py:119-127)
    int vec;                                 // every pointer 16-byte aligned (w16: 8) -> float4 path
};
struct ZeroJob { float* p; unsigned n; };      // one block per job
int launch_zero_multi(const ZeroJob* dev_jobs, int njobs, hipStream_t st);
struct TransposeJob { const uint16_t* src; uint16_t* dst; int R, C; unsigned blk0; int tiles_r; };   // dst [C, R] = src [R, C]^T, R and C multiples of 8
int launch_sgd_multi(const SgdSeg* dev_segs, int nseg, unsigned blocks, float lr_pre, float lr_scr, float mu, float wd, int first, int dtype,
                     hipStream_t st);
int launch_transpose16_multi(const TransposeJob* dev_jobs, int njobs, unsigned blocks, hipStream_t st);

// ---- multi-scale / flip evaluator data movement (evaluator.hip) ---------------------------------------------------------------
int launch_eval_make_crops(const float* img, float* crops, int C, int height, int width, int crop, int stride, int h_grids, int w_grids,
                           int flip, const float* pad3, hipStream_t st);
int launch_eval_accumulate(const float* outs, float* outputs, int K, int height, int width, int ph, int pw, int crop, int stride, int h_grids,
                           int w_grids, int flip, hipStream_t st);
int launch_eval_resize(const float* src, float* dst, int P, int Hi, int Wi, int Ho, int Wo, int accumulate, hipStream_t st);

}  // namespace lseg

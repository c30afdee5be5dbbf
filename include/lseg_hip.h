/*
 * lseg_hip.h -- C ABI of the MI355X-native LSeg forward engine (liblseg_hip.so).
 *
 * Drop-in boundary for ONE hot path of isl-org/lang-seg: LSegNet.forward()
 * (modules/models/lseg_net.py:160-205 and everything it calls).  The reference
 * has no FFI layer -- the path sits behind a Python class API -- so these
 * entry points are what a ctypes binding inside the reference's
 * modules/models/lseg_net.py would call.  Every function
 * cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C: pointers, sizes, ints.  No torch / C++ types cross the boundary.
 *   - every pointer named dev_* / d_* is DEVICE memory (HBM) owned by the caller;
 *     host pointers are named host_* .
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is stream-ordered, nothing synchronises the host unless documented.
 *   - one handle per device, not thread-safe per handle (the reference runs one
 *     Python thread per GPU replica: additional_utils/models.py:229-238).
 *   - return value: LSEG_OK (0) or a negative lseg_status; lseg_last_error()
 *     gives a message.  No C++ exception crosses the boundary.
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point returns LSEG_ERR_NO_DEVICE.
 */
#ifndef LSEG_HIP_H
#define LSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSEG_ABI_VERSION 1

typedef enum {
    LSEG_OK = 0,
    LSEG_ERR_INVALID = -1,     /* bad argument / shape */
    LSEG_ERR_NO_DEVICE = -2,   /* no HIP device / wrong arch */
    LSEG_ERR_HIP = -3,         /* a HIP runtime call failed */
    LSEG_ERR_STATE = -4,       /* call order (params not finalised, tokens not set...) */
    LSEG_ERR_UNSUPPORTED = -5, /* configuration outside what the engine implements */
    LSEG_ERR_MISSING_PARAM = -6
} lseg_status;

typedef enum { LSEG_F32 = 0, LSEG_F16 = 1, LSEG_BF16 = 2, LSEG_I64 = 3,
               LSEG_F16_SPLIT = 4   /* lseg_config.image_dtype only: split-precision validation mode, see below */
} lseg_dtype;

/* Resample op that follows the 1x1 conv of act_postprocessK
 * (modules/models/lseg_vit.py:446-523 / 315-396). */
typedef enum { LSEG_RS_IDENTITY = 0, LSEG_RS_CONVT = 1, LSEG_RS_CONV_S2 = 2 } lseg_resample;

/* Shape description of one network variant; mirrors the constants hard-coded at
 * lseg_net.py:119-123,142-146, lseg_vit.py:221-272, lseg_blocks.py:24-52 and the
 * [3P] timm / CLIP model dims.  Filled by the Python shim from `--backbone`. */
typedef struct {
    int32_t abi_version;      /* = LSEG_ABI_VERSION */
    /* image tower (timm VisionTransformer) */
    int32_t patch;            /* 16 | 32 */
    int32_t dim;              /* 1024 | 768 */
    int32_t depth;            /* 24 | 12 */
    int32_t heads;            /* 16 | 12 ; head_dim must be 64 */
    int32_t hooks[4];         /* lseg_net.py:119-123 */
    int32_t pos_grid;         /* pretrained pos-embed grid side (384/patch) */
    /* reassemble + DPT head */
    int32_t reassemble_ch[4]; /* channels after the 1x1 conv */
    int32_t resample_kind[4]; /* lseg_resample */
    int32_t resample_k[4];    /* ConvTranspose kernel=stride, or 3 for conv_s2 */
    int32_t features;         /* --num_features (256) */
    int32_t out_c;            /* 512 (768 for clipRN50x16_vitl16_384) */
    int32_t arch_option;      /* 0 | 1 (bottleneck_block) | 2 (depthwise_block), lseg_net.py:148-154 */
    int32_t block_depth;
    int32_t activation;       /* 0 relu, 1 lrelu, 2 tanh (lseg_net.py:47-52) */
    /* CLIP text tower */
    int32_t text_vocab, text_ctx, text_width, text_heads, text_layers;
    /* execution plan */
    int32_t img_h, img_w;     /* input size, multiples of patch (and such that the reassemble pyramid is a x2 ladder) */
    int32_t max_batch;        /* workspace is sized for this many images per call */
    int32_t max_labels;       /* workspace is sized for this many labels (K) */
    int32_t image_dtype;      /* MFMA operand type of the image tower: LSEG_BF16 (default), LSEG_F16, or LSEG_F16_SPLIT = every
                               * 16-bit operand as a (hi, lo) fp16 pair, three MFMA products per K-step (~21 mantissa bits: the
                               * reference's tower is fp32).  ~1/4 of the bf16 rate; a validation mode: its masks equal the
                               * reference's except at fp16-ulp ties of the reference's own fp16 logits.  Inference only. */
    int32_t flags;            /* bit 0: run the text tower on all text_ctx positions (reference schedule)
                               * instead of the exact causal truncation to max(EOT)+1 positions
                               * bit 1 (training): keep the gradient of the fp16 correlation in bf16 / fp32 instead of reproducing the
                               * reference's HALF-precision backward of `logit_scale * image_features.half() @ text_features.t()`
                               * (lseg_net.py:194): there d(logits), dA = d(logits) @ text and logit_scale * dA are fp16 tensors, and with
                               * d(logits) ~ 1 / (valid pixels) ~ 1e-6 they sit in fp16's subnormal range (step 6e-8) -- softmax
                               * probabilities below ~N * 3e-8 flush to zero.  Default (bit clear) = the reference's arithmetic.
                               * bit 2: batch-invariant schedule -- no split-K at small batches, so every GEMM accumulates K in one order
                               * whatever the batch size and image b of a batch equals the same image run alone to fp32 round-off
                               * (default: attn.proj / mlp.fc2 / the deep 3x3 convs split their contraction when B <= ~6)
                               * bit 3 (training): deterministic reductions -- the bias-gradient and BatchNorm column sums write partial
                               * rows summed in a fixed order instead of fp32 atomics: two runs of the same step on the same inputs
                               * give BIT-identical gradients (what torch.use_deterministic_algorithms buys the reference's
                               * training loop, modules/lsegmentation_module.py:66-81); one small extra launch per sum, no measurable cost per step (the
                               * Python front sets it by default: lseg_hip/engine.py, LSEG_DETERMINISTIC=0 clears it) */
} lseg_config;

typedef struct lseg_engine* lseg_handle;

/* ---- life cycle ------------------------------------------------------------------
 * replaces: LSegNet.__init__ / LSeg.__init__ (lseg_net.py:104-158,208-226) as far as
 * device-side state is concerned (module construction itself stays in Python). */
int lseg_abi_version(void);
int lseg_device_count(void);
int lseg_create(const lseg_config* cfg, int device, lseg_handle* out);
int lseg_destroy(lseg_handle h);
const char* lseg_last_error(lseg_handle h);   /* h may be NULL: last error of this thread */

/* ---- parameters ------------------------------------------------------------------
 * replaces: nn.Module.load_state_dict / BaseModel.load (lseg_net.py:81-92).  `key` is a
 * state-dict key relative to `net.`, e.g.
 * "pretrained.model.blocks.3.attn.qkv.weight".  The engine converts/repacks into its
 * own HBM copies (bf16/fp16 tiles, BN folded into the conv weights, ConvTranspose
 * re-laid as a GEMM) during lseg_finalize_params; the caller's tensor is only read.
 * Re-bind + finalize again after the weights change (e.g. after an optimizer step). */
int lseg_bind_param(lseg_handle h, const char* key, const void* dev_ptr, int dtype,
                    const int64_t* shape, int ndim);
int lseg_finalize_params(lseg_handle h, void* stream);

/* ---- text --------------------------------------------------------------------------
 * replaces: `text = clip.tokenize(labels)` being handed to
 * `clip_pretrained.encode_text(text)` (lseg_net.py:158,163-164,181-183) and the fp16
 * L2 normalisation (lseg_net.py:192).  Tokens are what clip.tokenize returns: int64
 * [K, ctx] on the HOST.  lseg_encode_text runs the 12-layer tower and leaves the
 * normalised fp16 features [K, out_c] in the engine.  lseg_forward re-runs it on every
 * call (reference semantics) unless text caching is switched on. */
int lseg_set_text_tokens(lseg_handle h, const int64_t* host_tokens, int K, int ctx);
/* Text features computed elsewhere -- the value of `self.clip_pretrained.encode_text(text)` (lseg_net.py:183), fp16 [K, out_c] in
 * device memory, not necessarily normalised -- instead of tokens: the engine applies the fp16 L2 normalisation of :192 and never runs
 * its text tower until the next lseg_set_text_tokens (label banks of an application, lseg_app.py:350-355). */
int lseg_set_text_features(lseg_handle h, const void* dev_feat_f16, int K, void* stream);
int lseg_encode_text(lseg_handle h, void* stream);
int lseg_set_text_cache(lseg_handle h, int enabled);   /* 0 (default) = re-encode per forward */
int lseg_get_text_features(lseg_handle h, void* dev_out_f16 /* [K,out_c] fp16 */, void* stream);
/* Zero-shot variant -- replaces: LSegNetZS.forward(x, class_info) (lseg_net_zs.py:177-214): every image brings
 * its OWN label set.  labels_per_image = k > 0: the K = B*k token rows are grouped per image, image b is
 * correlated with rows [b*k, (b+1)*k) only and lseg_forward writes [B,k,img_h,img_w]; 0 (default) = one
 * label set shared by the batch.  Takes effect at the next lseg_forward (which must be called with B = K/k). */
int lseg_set_text_grouping(lseg_handle h, int labels_per_image);

/* ---- forward -----------------------------------------------------------------------
 * replaces: LSeg.forward(x, labelset) (lseg_net.py:160-205).
 *   dev_x          fp32 [B,3,img_h,img_w] NCHW, normalised like lseg_module.py:37-50
 *   dev_logits_out fp32 [B,K,img_h,img_w], caller-allocated, freshly written (the caller
 *                  may mutate it afterwards: encoding_models.py:138).  May be NULL.
 *   dev_argmax_out uint8 [B,img_h,img_w]: the mask = argmax over the labels of the output logits (first maximum wins, like
 *                  torch.max(pred, 1) at lsegmentation_module.py:114-117), computed from the low-resolution logits through the
 *                  x2 bilinear on the fly; optional, may be NULL.  With dev_logits_out == NULL the full-resolution logits
 *                  (138 MB per image at K = 150) are never materialised.  K <= 256. */
int lseg_forward(lseg_handle h, const float* dev_x, int B, float* dev_logits_out,
                 uint8_t* dev_argmax_out, void* stream);

/* The metric / loss step after the path, without the full-resolution logits: statistics of the LAST lseg_forward's output against a
 * target mask -- replaces batch_pix_accuracy + batch_intersection_union on `pred` (lsegmentation_module.py:49-50,59-60) and the value
 * of the criterion (:72).  dev_target int64 [B,img_h,img_w]; dev_counts / dev_nll as in lseg_op_seg_stats. */
int lseg_forward_stats(lseg_handle h, const int64_t* dev_target, int ignore_index, int64_t* dev_counts, double* dev_nll, void* stream);

/* Intermediate taps for parity tests (names: "act1".."act4", "layer1".."layer4",
 * "rn1".."rn4", "path1".."path4", "image_features", "lowres").  Copies the tensor in the oracle's layout
 * ([B,N,D] / NCHW fp32) into dev_out; *n_elems gets the element count. */
int lseg_get_intermediate(lseg_handle h, const char* name, float* dev_out, size_t cap_elems,
                          size_t* n_elems, void* stream);
/* Debug mode keeps fp32 snapshots of the 4 hooked block outputs ("act1".."act4", the
 * reference's forward hooks at lseg_vit.py:12-16,421-424) during lseg_forward. */
int lseg_set_debug(lseg_handle h, int enabled);

/* ---- measurement -------------------------------------------------------------------
 * Per-kernel-family HIP-event timing on the engine's stream (bench.py roofline leg).
 * family names, in mask-bit order: "forward" (bit 0), "mlp_fc1", "mlp_fc2", "attn_proj", "attn_qkv", "attention", "layernorm" (the
 * ViT block's kernels: timm vision_transformer.py Block, invoked from lseg_vit.py:196-197).  lseg_set_profiling(h, 1) = forward +
 * mlp_fc1; any other non-zero value is a bit mask over the families; 0 switches the events off.  Events for 64 forwards are created
 * by the call, outside any timed region.  flops_per_launch = 2MNK of the family's GEMM (4 N^2 D per image for attention). */
int lseg_set_profiling(lseg_handle h, int enabled);
/* Range check of the image tower's 16-bit activations (every 16-bit activation buffer of the plan, as the forwards so far left them):
 * host_out4[0] = non-finite values (fp16 operands overflow at 65504 -- the reference's tower is fp32, lseg_vit.py:196-197, and cannot),
 * [1] = finite values with |x| >= 2^15, [2] = the largest finite |x| (fp32 bit pattern), [3] = elements scanned.  Synchronises the
 * stream.  The Python mirror runs it once after every (re)pack of an fp16 engine and falls back to bf16 loudly when [0] > 0. */
int lseg_check_range(lseg_handle h, uint64_t* host_out4, void* stream);
/* The always-on companion of lseg_check_range: every inference lseg_forward raises a device flag when the head feature map carries an inf / NaN
 * (any non-finite value of the 16-bit image tower ends up there) and copies it to pinned host memory behind the forward.  This call never
 * synchronises: it returns 1 once a COMPLETED forward was flagged (sticky until `reset`), else 0 -- i.e. an overflow on image n is seen at the
 * latest when image n+1 is submitted.  The Python mirror falls back to bf16 operands loudly on it (fp16 saturates at 65504; the reference's
 * tower is fp32, lseg_vit.py:196-197). */
int lseg_overflow_seen(lseg_handle h, int reset);
int lseg_get_profile(lseg_handle h, const char* family, double* total_ms, int64_t* launches,
                     double* flops_per_launch);

/* ---- single-operator entry points (unit parity tests call these through the ABI) --
 * All tensors are device pointers; dtype codes are lseg_dtype. */

/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias) (+ residual); A,W: bf16 or fp16 (`ab_dtype`),
 * bias fp32 [N] or NULL, residual fp32 [M,N] or NULL, C: out_dtype.  act: 0 none,
 * 1 GELU(erf), 2 QuickGELU, 3 ReLU.   (timm Linear / CLIP Linear / 1x1 conv) */
int lseg_op_gemm(const void* d_A, const void* d_W, const float* d_bias, const float* d_residual,
                 void* d_C, int M, int N, int K, int ab_dtype, int out_dtype, int act, void* stream);
/* The four GEMM forms of a timm Block (lseg_vit.py:196-197 -> [3P] timm Block / Attention / Mlp) on the engine's specialised epilogues,
 * standalone: kind 0  C = T(A W^T + b)  (T = ab_dtype);  1  C = T(gelu(A W^T + b))  (mlp.fc1);  2  C += A W^T + b on an fp32 C (the residual
 * stream: attn.proj, mlp.fc2; d_Cq is read and written);  3  the qkv Linear with timm's reshape/permute folded into the store: d_Cq = q and
 * d_Ck = k as [B*heads, npad, 64], d_Cv = v TRANSPOSED [B*heads, 64, npad] (N = 3*heads*64, M = B*ntok, pad rows untouched).
 * max_grid > 0 caps the persistent grid (tools/epilogue_table.py).  d_bias fp32 [N] is required. */
int lseg_op_gemm_vit(const void* d_A, const void* d_W, const float* d_bias, void* d_Cq, void* d_Ck, void* d_Cv, int M, int N, int K,
                     int ab_dtype, int kind, int ntok, int npad, int max_grid, void* stream);
/* The residual form alone -- d_C (fp32 [rows_alloc, N], rows_alloc >= M) += A W^T + bias, in place: attn.proj / mlp.fc2 of a timm Block
 * (lseg_vit.py:196-197) -- with the caller stating how many rows of d_A and d_C are ALLOCATED.  When they reach the next multiple of 256
 * the launch takes the hand-scheduled 256 x 128 kernel (csrc/gemm_asm.hip: whole tiles, no masks; rows >= M of d_C are overwritten with
 * values nobody should read); impl = 0 forces the generic kernel family, 1 requires the hand-scheduled one (LSEG_ERR_UNSUPPORTED when the
 * shape does not qualify), -1 lets the launcher choose like the engine does.  max_grid > 0 caps the persistent grid. */
int lseg_op_gemm_res32(const void* d_A, const void* d_W, const float* d_bias, float* d_C, int M, int N, int K, int rows_alloc,
                       int ab_dtype, int impl, int max_grid, void* stream);
/* Column reductions of the training step, standalone (the bias gradients of every Linear / conv and the BatchNorm batch statistics of
 * the DPT residual units: autograd of nn.Linear / nn.Conv2d / nn.BatchNorm2d under modules/lsegmentation_module.py:66-81).
 *   lseg_op_colsum:       d_out[c] (+)= sum_r in[r, c], in 16-bit [R, ld] row-major, C <= ld columns, fp32 out
 *   lseg_op_bn_stats:     d_stats[c] = sum x, d_stats[C + c] = sum x^2 over a padded NHWC map [B, H+2, W+2, C] (zero border)
 *   lseg_op_bn_bwd_stats: d_bstats[c] = sum dy, d_bstats[C + c] = sum dy (x - mean_c) rstd_c, mean / rstd from d_stats over `count` = B H W
 * d_det_ws (optional, det_cap_floats floats): the DETERMINISTIC form (lseg_config.flags bit 3) -- one partial row per row block, summed in
 * a fixed order -- instead of fp32 atomics.  Same values up to summation order; two runs of the deterministic form are bit-identical. */
int lseg_op_colsum(const void* d_in, int dtype, float* d_out, int R, int C, int ld, int accumulate, float* d_det_ws, size_t det_cap_floats,
                   void* stream);
int lseg_op_bn_stats(const void* d_x_padded, float* d_stats, int B, int H, int W, int C, int dtype, float* d_det_ws, size_t det_cap_floats,
                     void* stream);
int lseg_op_bn_bwd_stats(const void* d_dy_padded, const void* d_x_padded, const float* d_stats, float* d_bstats, int B, int H, int W, int C,
                         float eps, int dtype, float* d_det_ws, size_t det_cap_floats, void* stream);
/* LayerNorm over the last dim: in fp32|fp16 [M,D] -> out bf16|fp16 [M,D] */
int lseg_op_layernorm(const void* d_in, int in_dtype, const float* d_gamma, const float* d_beta,
                      void* d_out, int out_dtype, int M, int D, float eps, void* stream);
/* softmax(Q K^T * scale [+causal]) V for head_dim 64.
 * q,k: [BH, Npad, 64]; vt: [BH, 64, Npad] (V transposed); out: [B, Ntok, H*64]. */
int lseg_op_attention(const void* d_q, const void* d_k, const void* d_vt, void* d_out,
                      int B, int H, int Ntok, int Npad, int dtype, int causal, float scale,
                      void* stream);
/* The same product for a q that already carries softmax scale * log2(e) (the inference engine folds it into the qkv Linear's epilogue,
 * one rounding): out = softmax2(Q' K^T) V with softmax2 in base 2, no mask.  d_lse2 (optional, fp32 [B*H, Npad]) receives log2 sum_k
 * 2^(q'.k).  Same layouts as lseg_op_attention. */
int lseg_op_attention_prescaled(const void* d_q, const void* d_k, const void* d_vt, void* d_out, float* d_lse2,
                                int B, int H, int Ntok, int Npad, int dtype, void* stream);
/* 3x3 conv, NHWC, zero-padded borders: in [B,H+2,W+2,Cin] (bf16, border = 0) ->
 * out [B,Ho+2,Wo+2,Cout] interior written; w_packed [Cout, 9*Cin] bf16 (tap-major).
 * stride 1|2, relu_in / relu_out flags, bias fp32 [Cout] or NULL, residual (same
 * geometry as out, bf16) or NULL. */
int lseg_op_conv3x3(const void* d_in, const void* d_w_packed, const float* d_bias,
                    const void* d_residual, void* d_out, int B, int H, int W, int Cin, int Cout,
                    int stride, int relu_in, int relu_out, void* stream);
/* bilinear x2, align_corners=True, NHWC bf16: in [B,H+2,W+2,C] (padded) -> out [B,2H,2W,C] */
int lseg_op_upsample2x_nhwc(const void* d_in, void* d_out, int B, int H, int W, int C, void* stream);
/* bilinear x2, align_corners=True, NCHW fp32 planes: in [P,H,W] -> out [P,2H,2W]
 * (scratch.output_conv, lseg_net.py:203,219-221) */
int lseg_op_upsample2x_planes(const float* d_in, float* d_out, int P, int H, int W, void* stream);
/* The tail of the commuted head (DESIGN.md par. 3.4) in one pass: label planes R at the quarter resolution, d_in_padded fp32 [B*K, H+2, W+2]
 * (1-pixel border), per-pixel scale s/||u|| d_scale fp32 [B, 2H, 2W] -> logits d_out fp32 [B, K, 4H, 4W] =
 * output_conv(fp16(scale * bilinear_x2(R))) (lseg_net.py:194-196 rounding, :203 upsample).  two_stage != 0 runs the two materialising
 * kernels instead (d_low_scratch fp32 [B,K,2H,2W]): same bits. */
int lseg_op_upsample4x_planes_scaled(const float* d_in_padded, const float* d_scale, float* d_out, int B, int K, int H, int W, int two_stage,
                                     float* d_low_scratch, void* stream);
/* pixel x text correlation (lseg_net.py:187-196): feat fp32 [M,C]; text fp16 [K,C]
 * (already L2-normalised); logits fp32 [B,K,P] with M = B*P.  Internally:
 * a = fp16(scale * fp16(feat/||feat||)); logits = fp16(a @ text^T). */
int lseg_op_correlation(const float* d_feat, const void* d_text_f16, float* d_logits,
                        int B, int P, int C, int K, float logit_scale, void* stream);

/* fused scratch.head1 + pixel normalisation (lseg_net.py:185-194, out_c = 512): x bf16 [M,F] pixels,
 * w bf16 [512,F], bias fp32 [512]  ->  a fp16 [M,512] = fp16(scale * fp16(v/||v||_2)), v = x w^T + bias
 * (the A operand of the correlation GEMM; the fp32 features are never materialised). */
int lseg_op_head_features(const void* d_x_bf16, const void* d_w_bf16, const float* d_bias, void* d_a_f16,
                          int M, int F, float logit_scale, void* stream);

/* Segmentation statistics of a score tensor against a target mask, on the device (one pass over the scores).
 * replaces: the host-side metric / loss step after LSeg.forward --
 *   batch_pix_accuracy + batch_intersection_union ([3P] encoding/utils/metrics.py; called at
 *   lsegmentation_module.py:49-50,59-60 and through SegmentationMetric.update at test_lseg.py:385-388), and the
 *   forward value of SegmentationLosses = nn.CrossEntropyLoss(ignore_index) ([3P] encoding/nn/loss.py; :72).
 * d_scores fp32 [B,K,H,W]; d_target int64 [B,H,W] (values -1 = unlabeled / 0..K-1);
 * d_counts int64 [2+3K] = {correct, labeled, area_inter[K], area_pred[K], area_lab[K]}  (union = pred + lab - inter);
 * d_nll double [2] = {sum over valid pixels of -log_softmax(scores)[target], number of valid pixels}. */
int lseg_op_seg_stats(const float* d_scores, const int64_t* d_target, int B, int K, int H, int W, int ignore_index,
                      int64_t* d_counts, double* d_nll, void* stream);
/* The same statistics (and / or the masks) straight from the LOW-resolution logits [B,K,h,w] the engine keeps before
 * scratch.output_conv (lseg_net.py:203): every pixel of the [B,2h,2w] grid reads them through the x2 bilinear (align_corners=True)
 * on the fly, so the metric / mask step needs no full-resolution logits.  d_target int64 [B,2h,2w] or NULL (then d_counts / d_nll
 * are not touched); d_argmax uint8 [B,2h,2w] or NULL. */
int lseg_op_seg_stats_lowres(const float* d_low, const int64_t* d_target, int B, int K, int h, int w, int ignore_index,
                             int64_t* d_counts, double* d_nll, uint8_t* d_argmax, void* stream);

/* The pixel x text correlation on the engine's commuted schedule (DESIGN.md par. 3.4), one dedicated kernel (csrc/corr.hip).
 * replaces: `logits_per_image = self.logit_scale * image_features.half() @ text_features.t()` (modules/models/lseg_net.py:194) together
 *   with the norm of lseg_net.py:190 -- as label planes R[b, k, p] = t_k . g_p on the quarter-resolution map g (the combined
 *   head1 o out_conv 1x1 conv output, padded NHWC fp16) and the five dot products of every 2x2 cell of g that the norm of the
 *   x2-up-sampled feature is made of; the scale / fp16 rounding / upsample kernels consume both.
 * d_g16pad fp16 [B, H+2, W+2, C] (C = 512); d_text16 fp16 [K, C] (K x (2C + 16) bytes must fit the 160 KB LDS: K <= 157);
 * d_planes fp32 [B, K, (H+2)(W+2)] -- interior pixels written; d_gram fp32 [B, H, W, 5] = {g.g, g.g(x+1), g.g(y+1), g.g(y+1,x+1),
 * g(x+1).g(y+1)} or NULL (planes only).  LSEG_ERR_UNSUPPORTED for other C / larger K (the engine then takes its generic GEMM). */
int lseg_op_corr_planes(const void* d_g16pad, const void* d_text16, float* d_planes, float* d_gram, int B, int K, int H, int W, int C,
                        void* stream);

/* Device side of the multi-scale / flip sliding-window evaluator that calls the forward (additional_utils/encoding_models.py:54-155
 * MultiEvalModule.forward, module_inference, pad_image, crop_image, flip_image; additional_utils/models.py:55-140).  Per scale:
 *   lseg_op_eval_make_crops  d_img fp32 [C,height,width] (the resized image, C <= 3) -> d_crops fp32 [(1+flip)*n, C, crop, crop], n =
 *                            h_grids*w_grids: crop (idh,idw) starts at (idh*stride, idw*stride), pixels beyond the image take pad[c]
 *                            (= -mean/std, :144-155); with flip the second half holds the mirrored twins (:135-137)
 *   lseg_op_eval_accumulate  d_outs fp32 [(1+flip)*n, K, crop, crop] (the engine's logits for that stack) -> d_map fp32 [K,height,width]:
 *                            out + flip(out_twin) summed over the covering boxes in (idh,idw) order, divided by their count, cropped
 *                            to the un-padded size (:100-121); ph, pw = size of the padded image the boxes are clipped to
 *   lseg_op_eval_resize      d_dst [P,Ho,Wo] (+)= bilinear(d_src [P,Hi,Wi]), align_corners=True (:78, :123: the image resize and
 *                            `scores += resize_image(outputs, h, w)`) */
int lseg_op_eval_make_crops(const float* d_img, float* d_crops, int C, int height, int width, int crop, int stride, int h_grids, int w_grids,
                            int flip, const float* host_pad3, void* stream);
int lseg_op_eval_accumulate(const float* d_outs, float* d_map, int K, int height, int width, int ph, int pw, int crop, int stride,
                            int h_grids, int w_grids, int flip, void* stream);
int lseg_op_eval_resize(const float* d_src, float* d_dst, int P, int Hi, int Wi, int Ho, int Wo, int accumulate, void* stream);

/* Backward of one Linear layer y = x W^T + b -- first brick of the training step (the reference gets
 * it from torch autograd under LSegmentationModule.training_step, lsegmentation_module.py:66-81).  bf16/fp16 operands,
 * fp32 accumulate, both GEMMs on the forward MFMA kernel with the contraction dimension transposed onto the fast axis:
 *   d_dx [M,K] (operand dtype)  = dY [M,N] . W [N,K]          (may be NULL)
 *   d_dw [N,K] fp32             = dY^T [N,M] . X [M,K]         (may be NULL)
 *   d_db [N]   fp32             = column sums of dY            (may be NULL)
 * N and K multiples of 64. */
int lseg_op_linear_backward(const void* d_dy, const void* d_x, const void* d_w, int ab_dtype, void* d_dx, float* d_dw,
                            float* d_db, int M, int N, int K, void* stream);

/* Backward of LayerNorm over the last dimension (timm norm1/norm2 on the fp32 residual stream; autograd in the
 * reference): d_dy [M,D] (fp32 / bf16 / fp16), d_x [M,D] fp32 (the forward input), d_gamma [D];
 * d_dx [M,D] fp32 (= or += when accumulate_dx: the residual stream's gradient), d_dgamma / d_dbeta [D] fp32. */
int lseg_op_layernorm_backward(const void* d_dy, int dy_dtype, const float* d_x, const float* d_gamma, float* d_dx,
                               float* d_dgamma, float* d_dbeta, int M, int D, float eps, int accumulate_dx, void* stream);

/* Backward of a stride-1 3x3 convolution in the padded-NHWC layout (the DPT head's convs, lseg_blocks.py:73-108,
 * 237-255; autograd in the reference), bf16 operands, built on the forward kernels:
 *   d_dx_pad [B,H+2,W+2,Cin] bf16, interior written (caller keeps the border zero) = conv3x3(dY, W flipped / channel-swapped)
 *   d_dw     [Cout, 9*Cin]  fp32 (tap-major, like d_w_packed)            = dY^T x (9 row-shifted copies of X)^T, one GEMM
 * d_dy_pad [B,H+2,W+2,Cout] bf16 with a ZERO border, d_x_pad the forward input.  Either output may be NULL. */
int lseg_op_conv3x3_backward(const void* d_dy_pad, const void* d_x_pad, const void* d_w_packed, void* d_dx_pad, float* d_dw,
                             int B, int H, int W, int Cin, int Cout, void* stream);

/* More backward bricks (autograd in the reference; each checked against torch autograd in tests/test_gpu_ops.py):
 *   gelu_backward           d_dx = d_dy * GELU'(d_pre), erf form (timm Mlp.act), bf16/fp16 [n]
 *   upsample2x_nhwc_backward  transpose of lseg_op_upsample2x_nhwc: d_dout [B,2H,2W,C] bf16 -> d_din_pad [B,H+2,W+2,C] interior
 *   softmax_ce_backward     d_dscores [B,K,H,W] fp32 = (softmax_k - 1[k = target]) / n_valid (0 at ignored pixels);
 *                           d_nll = the double[2] written by lseg_op_seg_stats (n_valid in d_nll[1]) */
int lseg_op_gelu_backward(const void* d_dy, const void* d_pre, void* d_dx, int64_t n, int dtype, void* stream);
int lseg_op_quickgelu_backward(const void* d_dy, const void* d_pre, void* d_dx, int64_t n, int dtype, void* stream);   /* CLIP: x*sigmoid(1.702x) */
int lseg_op_upsample2x_nhwc_backward(const void* d_dout, void* d_din_pad, int B, int H, int W, int C, void* stream);
int lseg_op_softmax_ce_backward(const float* d_scores, const int64_t* d_target, float* d_dscores, int B, int K, int H, int W,
                                int ignore_index, const double* d_nll, void* stream);

/* Backward of lseg_op_attention (softmax(Q K^T * scale) V, head_dim 64, no mask); the reference gets it from autograd through
 * [3P] timm Attention.forward (lseg_vit.py:196-197).  Flash-style recomputation (csrc/attention_bwd2.hip), no atomics:
 * d_q,d_k [BH,Npad,64], d_vt [BH,64,Npad], d_o / d_do [B,Ntok,H*64] (forward layouts, bf16/fp16); d_lse2 [BH,Npad] fp32 = log2 of
 * each query row's sum_k exp2(s_k * scale * log2e) (the forward's running statistics) -- a dQ kernel and a
 * dK/dV kernel in the forward kernel's MFMA / direct-to-LDS structure, writing d(qkv Linear output) [B*Ntok, 3*H*64] (bf16/fp16)
 * directly.  d_ws: lseg_op_attention_backward_ws_bytes(B, H, Npad) bytes of device scratch, or NULL (then allocated per call). */
size_t lseg_op_attention_backward_ws_bytes(int B, int H, int Npad);
int lseg_op_attention_backward_qkv(const void* d_q, const void* d_k, const void* d_vt, const void* d_o, const void* d_do,
                                   const float* d_lse2, void* d_dqkv, void* d_ws, int B, int H, int Ntok, int Npad, int dtype,
                                   float scale, void* stream);

/* Train-mode BatchNorm2d on the padded-NHWC bf16 maps (ResidualConvUnit_custom bn1/bn2 under net.train(),
 * lseg_blocks.py:276-283; the reference runs SyncBatchNorm, i.e. d_stats / the backward sums are what a multi-GPU step
 * all-reduces).  Maps are [B,H+2,W+2,C] with a zero border; statistics over the B*H*W image pixels, biased variance.
 *   forward : d_stats [2C] fp32 = {sum x, sum x^2}; d_y_pad (may be NULL: statistics only) = gamma*(x-mean)*rstd + beta
 *   backward: d_dx_pad = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)); d_dgamma_dbeta [2C] = {dbeta = sum dy, dgamma = sum dy*xhat}
 * relu_backward: d_dx = d_dy where d_x > 0 (bf16/fp16, n elements). */
int lseg_op_bn_train_forward(const void* d_x_pad, void* d_y_pad, float* d_stats, const float* d_gamma, const float* d_beta,
                             int B, int H, int W, int C, float eps, void* stream);
int lseg_op_bn_train_backward(const void* d_dy_pad, const void* d_x_pad, const float* d_stats, const float* d_gamma, void* d_dx_pad,
                              float* d_dgamma_dbeta, int B, int H, int W, int C, float eps, void* stream);
int lseg_op_relu_backward(const void* d_dy, const void* d_x, void* d_dx, int64_t n, void* stream);

/* Head-side backward bricks (lseg_net.py:185-203 under autograd):
 *   upsample2x_planes_backward_rows  d_dout [B,K,2H,2W] fp32 (d of the output logits) -> d_rows [B*H*W, ldk] bf16/fp16, columns
 *                                    0..K-1 (the rest pre-zeroed by the caller): transpose of output_conv's x2 bilinear, already in
 *                                    the row layout the correlation backward (lseg_op_linear_backward with N = ldk) consumes
 *   l2norm_scale_backward            d_dx [M,C] = (scale/||x||) (d_da - xh (xh . d_da)), xh = x/||x||; x fp32, da/dx bf16|fp16 */
int lseg_op_upsample2x_planes_backward_rows(const float* d_dout, void* d_rows, int B, int K, int H, int W, int ldk, int out_dtype,
                                            void* stream);
int lseg_op_l2norm_scale_backward(const void* d_da, int da_dtype, const float* d_x, void* d_dx, int dx_dtype, int M, int C, float scale,
                                  void* stream);
/* Fused backward of the loss the reference takes on the full-resolution logits -- CrossEntropyLoss(ignore_index)(output_conv(low)),
 * lsegmentation_module.py:72 on lseg_net.py:203 -- from the LOW-resolution logits d_low fp32 [B,K,h,w] and the target mask int64
 * [B,2h,2w]: the x2 bilinear, the softmax and the bilinear's transpose in registers, so the [B,K,2h,2w] logits and their gradient
 * (2 x 138 MB per image at K = 150, 480x480) never exist.  Outputs: d_nll double [2] = {sum of -log_softmax[target] over the valid
 * pixels, number of valid pixels} (the loss is their ratio), d_rows [B*h*w, ldk] bf16|fp16 = d loss / d low in the row layout the
 * correlation backward consumes (all ldk columns written, zeros beyond K; ldk % 8 == 0), d_lse_ws fp32 [B*2h*2w] scratch. */
int lseg_op_upsample_ce_backward_rows(const float* d_low, const int64_t* d_target, int B, int K, int h, int w, int ignore_index,
                                      double* d_nll, float* d_lse_ws, void* d_rows, int ldk, int out_dtype, void* stream);

/* ---- training step ------------------------------------------------------------------------------------------------------------
 * replaces: LSegmentationModule.training_step (modules/lsegmentation_module.py:66-81) -- `out = self(img)` in train() mode,
 * `loss = criterion(out, target)` (SegmentationLosses = CrossEntropyLoss(ignore_index), [3P] encoding/nn/loss.py), autograd's
 * backward -- plus what the Trainer does around it: DistributedDataParallel's bucketed gradient all-reduce and SyncBatchNorm
 * (utils.py:20-22,34) through two callbacks, and SGD with the two learning-rate groups of configure_optimizers (:119-127,165-171).
 *
 *   lseg_set_train(h, 1)   net.train(): lseg_forward keeps the activations the backward needs, the refinenets' BatchNorm uses batch
 *                          statistics and updates running_mean / running_var IN the caller's bound tensors (momentum 0.1).  bf16 only.
 *   lseg_bind_grad         where the gradient of parameter `key` is written: fp32, same shape/layout as the bound parameter (the
 *                          caller's .grad tensor, typically a view into a flat bucket).  Unbound parameters get engine-owned buffers
 *                          (lseg_grad_ptr).  Trainable = pretrained.* and scratch.* tensors the forward touches; the CLIP text tower
 *                          is frozen (it is in no optimizer group of the reference) and its features are constants of the step.
 *   lseg_backward          after a train-mode lseg_forward: either dev_dlogits fp32 [B,K,H,W] (autograd hands it over), or
 *                          dev_target int64 [B,H,W] (then the loss is the mean CE over pixels != ignore_index and dev_loss, if not
 *                          NULL, receives double[2] = {sum of -log p[target], number of valid pixels}).  accumulate != 0 adds to the
 *                          gradient buffers (accumulate_grad_batches), 0 overwrites them.
 *   lseg_train_loss        the criterion's VALUE on the last train-mode forward (`loss = self.criterion(out, target)`, :72) without the
 *                          backward: dev_loss double[2] as above, dev_counts (or NULL) int64[2] = {correct, labeled} of the arg-max
 *                          mask on the valid pixels (`train_accuracy`, :76-79).  A following lseg_backward / lseg_backward_scaled on
 *                          the same target reuses its per-pixel log-sum-exp.
 *   lseg_backward_scaled   lseg_backward(target) whose d(logits) is multiplied by *dev_grad_scale (a float in device memory, read by
 *                          the kernel: no host synchronisation) -- the d(loss) autograd hands to `loss.backward()` (:81 via Lightning).
 *   lseg_grad_bucket       gradients complete in buckets: 0 = DPT head + reassemble, 1+j = ViT block depth-1-j (+ the readout hooked
 *                          on it), the last one also patch_embed / cls_token / pos_embed.  The bucket callback fires on the host right
 *                          after the last kernel of a bucket has been ENQUEUED on `stream` (record an event there and launch the
 *                          all-reduce on a side stream: it overlaps the remaining backward GEMMs).
 *   lseg_set_bn_sync       SyncBatchNorm: `fn(user, dev_ptr, n, stream)` must sum dev_ptr[0..n) over the ranks in place, ordered on
 *                          `stream`; called for the 2C batch sums of every BatchNorm (forward) and the 2C gradient sums (backward).
 *                          world_size = number of ranks (divides the sums).  NULL / 1 = per-GPU statistics.
 *   lseg_sgd_step          w -= lr * (mu * m + g + wd * w) on the bound fp32 parameters with the engine's momentum buffers, then
 *                          re-packs the MFMA operand copies.
 *   lseg_sgd_momentum      device pointer of a parameter's momentum buffer (torch.optim.SGD's state['momentum_buffer'], what
 *                          Lightning checkpoints under 'optimizer_states'); lseg_sgd_mark_initialized(h, 1) after restoring them makes
 *                          the next step a regular one (the first step of SGD copies the gradient into the buffer instead). */
typedef void (*lseg_reduce_cb)(void* user, void* dev_ptr, int64_t n_floats, void* stream);
typedef void (*lseg_bucket_cb)(void* user, int bucket, void* stream);
int lseg_set_train(lseg_handle h, int enabled);
int lseg_bind_grad(lseg_handle h, const char* key, float* dev_grad);
int lseg_grad_ptr(lseg_handle h, const char* key, float** dev_out, size_t* n_elems);
int lseg_grad_bucket(lseg_handle h, const char* key);          /* bucket index, or -1 if `key` is not a trainable parameter */
int lseg_num_grad_buckets(lseg_handle h);
int lseg_backward(lseg_handle h, const float* dev_dlogits, const int64_t* dev_target, int ignore_index, int accumulate,
                  double* dev_loss, void* stream);
int lseg_train_loss(lseg_handle h, const int64_t* dev_target, int ignore_index, double* dev_loss, int64_t* dev_counts, void* stream);
int lseg_backward_scaled(lseg_handle h, const int64_t* dev_target, int ignore_index, int accumulate, const float* dev_grad_scale, void* stream);
int lseg_sgd_momentum(lseg_handle h, const char* key, float** dev_out, size_t* n_elems);
int lseg_sgd_mark_initialized(lseg_handle h, int initialized);
int lseg_set_bn_sync(lseg_handle h, lseg_reduce_cb fn, void* user, int world_size);
int lseg_set_bucket_callback(lseg_handle h, lseg_bucket_cb fn, void* user);
int lseg_sgd_step(lseg_handle h, float lr_pretrained, float lr_scratch, float momentum, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSEG_HIP_H */

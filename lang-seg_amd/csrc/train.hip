// train.hip -- the training step of the LSeg path on one MI355X: train-mode forward (activations saved, BatchNorm on batch
// statistics), backward of everything LSegmentationModule.training_step differentiates (modules/lsegmentation_module.py:66-81:
// `out = self(img)`; `loss = criterion(out, target)`; autograd), gradients written straight into caller-provided fp32 buffers
// laid out like the reference's parameters, per-bucket "gradients enqueued" notifications for the RCCL all-reduce, fused SGD.
//
// What the reference's autograd would traverse and what runs here:
//   output_conv x2 bilinear, CrossEntropyLoss(ignore_index)      seg_stats (through the bilinear) + upsample_ce_bwd_rows on the low-res logits
//   fp16 correlation, L2-norm, head1 (lseg_net.py:185-196)       two GEMMs + l2norm_scale_backward (fp16 roundings = identity)
//   4 x FeatureFusionBlock_custom (lseg_blocks.py:337-358)       out_conv GEMMs, x2 upsample transpose, RCU: conv dgrad = forward
//     with ResidualConvUnit_custom in train() mode (:265-288)    conv on flipped weights, wgrad = one GEMM on transposed operands,
//                                                                BatchNorm backward on batch statistics (SyncBatchNorm exchange hook)
//   layerN_rn, act_postprocess (lseg_vit.py:446-523)             conv / strided conv (dilated dY) / ConvTranspose-as-GEMM backward
//   ProjectReadout (lseg_vit.py:79-90)                           GELU', Linear backward, cat/cls scatter-add into the block gradient
//   24 x timm Block (invoked lseg_vit.py:196-197)                Linear dgrad/wgrad GEMMs, GELU', LayerNorm backward, flash attention
//                                                                backward with the forward's saved log-sum-exp
//   patch_embed, cls_token, pos_embed resize (:166-193)          wgrad GEMM on the saved im2col, column sums, bilinear transpose
// The CLIP text tower is frozen (it is in no optimizer group, lsegmentation_module.py:119-127): its features are constants here;
// the reference back-propagates into it and discards the result.
//
// Gradients travel between kernels as bf16 (the per-logit gradient of a mean over 1.8 M pixels is ~1e-7: fp16 would flush it).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "engine.h"

namespace lseg {

#define TRY(expr) do { int _r = (expr); if (_r != 0) return _r; } while (0)
#define TALLOC(ptr, type, count)                                                                  \
    do {                                                                                          \
        ptr = (type*)dalloc((size_t)(count) * sizeof(type));                                       \
        if (!ptr) return set_error(LSEG_ERR_HIP, "hipMalloc of %zu bytes failed (" #ptr ")",      \
                                   (size_t)(count) * sizeof(type));                               \
    } while (0)

static inline size_t up64(size_t v) { return (v + 63) / 64 * 64; }

int Engine::set_train(bool on) {
    if (on) {
        if (img_dt_ != DT_BF16) return set_error(LSEG_ERR_UNSUPPORTED, "training needs bf16 image-tower operands (fp16 gradients underflow)");
        if (cfg.arch_option != 0) return set_error(LSEG_ERR_UNSUPPORTED, "training with arch_option %d head blocks is not implemented", cfg.arch_option);
        TRY(train_alloc());
        if (finalized_) TRY(finalize_train(nullptr));
    }
    train_mode = on;
    train_fwd_valid_ = false;
    return 0;
}

int Engine::bucket_of(const std::string& key) const {
    // bucket 0: DPT head + reassemble (their gradients are complete first); bucket 1 + j: ViT block depth-1-j together with
    // the ProjectReadout hooked on it; the last bucket (block 0) also carries patch_embed / cls_token / pos_embed.
    const std::string bp = "pretrained.model.blocks.";
    if (key.compare(0, bp.size(), bp) == 0) return 1 + (cfg.depth - 1 - atoi(key.c_str() + bp.size()));
    const std::string ap = "pretrained.act_postprocess";
    if (key.compare(0, ap.size(), ap) == 0) {
        const int l = key[ap.size()] - '1';
        if (key.compare(ap.size() + 1, 3, ".0.") == 0 && l >= 0 && l < 4) return 1 + (cfg.depth - 1 - cfg.hooks[l]);
        return 0;
    }
    if (key.compare(0, 8, "scratch.") == 0) return 0;
    if (key.compare(0, 17, "pretrained.model.") == 0) return cfg.depth;
    return -1;
}

int Engine::bind_grad(const char* key, float* p) {
    if (!key || !p) return set_error(LSEG_ERR_INVALID, "bind_grad: NULL argument");
    auto it = bound_.find(key);
    if (it == bound_.end()) return set_error(LSEG_ERR_MISSING_PARAM, "bind_grad: parameter '%s' was never bound", key);
    GradSlot& s = grads_[key];
    s.ptr = p; s.n = it->second.numel(); s.bound = true;
    sgd_dirty_ = true;
    zero_bwd_.ready = false; zero_bwd_.host.clear();
    return 0;
}

float* Engine::grad(const std::string& key, size_t n) {
    GradSlot& s = grads_[key];
    if (!s.ptr) { s.ptr = (float*)dalloc(n * sizeof(float)); s.n = n; sgd_dirty_ = true; }
    if (s.n != n) { set_error(LSEG_ERR_INVALID, "gradient of '%s' has %zu elements, expected %zu", key.c_str(), s.n, n); return nullptr; }
    return s.ptr;
}

int Engine::grad_ptr(const char* key, float** out, size_t* n) {
    auto it = grads_.find(key ? key : "");
    if (it == grads_.end() || !it->second.ptr) return set_error(LSEG_ERR_STATE, "no gradient for '%s' (run lseg_backward first)", key ? key : "");
    if (out) *out = it->second.ptr;
    if (n) *n = it->second.n;
    return 0;
}

// ---- one-launch zeroing of the atomically accumulated buffers ---------------------------------------------------------------------
bool Engine::zero_note(ZeroSet& z, float* p, size_t n) {
    if (z.ready) return true;
    ZeroJob j; j.p = p; j.n = (unsigned)n;
    z.host.push_back(j);
    return false;
}
int Engine::zero_begin(ZeroSet& z, hipStream_t st) {
    if (!z.ready) { z.host.clear(); return 0; }
    return launch_zero_multi(z.dev, z.n, st);
}
int Engine::zero_end(ZeroSet& z) {
    if (z.ready || z.host.empty()) return 0;
    z.dev = (ZeroJob*)dalloc(z.host.size() * sizeof(ZeroJob));
    if (!z.dev) return set_error(LSEG_ERR_HIP, "out of device memory (zero table)");
    LSEG_HIP_TRY(hipMemcpy(z.dev, z.host.data(), z.host.size() * sizeof(ZeroJob), hipMemcpyHostToDevice));
    z.n = (int)z.host.size();
    z.ready = true;
    return 0;
}
// db[c] (+)= column sums of dy: the bias gradient of a Linear / 1x1 conv
int Engine::bias_sum(const uint16_t* dy, float* db, int R, int C, int ld, int acc, hipStream_t st) {
    if (acc) return launch_colsum16(dy, img_dt_, db, R, C, ld, st, 1, ws_det_, ws_det_n_);
    if (ws_det_) return launch_colsum16(dy, img_dt_, db, R, C, ld, st, 0, ws_det_, ws_det_n_);       // overwrites: nothing to pre-zero
    return launch_colsum16(dy, img_dt_, db, R, C, ld, st, zero_note(zero_bwd_, db, (size_t)C) ? 1 : 0);
}

// ---- workspace of the training step (sized for cfg.max_batch images) -------------------------------------------------------
int Engine::train_alloc() {
    if (train_alloc_) return 0;
    LSEG_HIP_TRY(hipSetDevice(device));
    const lseg_config& c = cfg;
    const size_t B = c.max_batch, D = c.dim, F = c.features, M = B * ntok_, Mr = B * np_, H = c.heads;
    sv_.resize(c.depth);
    for (int i = 0; i < c.depth; ++i) {
        BlockSave& s = sv_[i];
        TALLOC(s.xin, float, M * D); TALLOC(s.xmid, float, M * D); TALLOC(s.lse, float, B * H * npad_);
        TALLOC(s.ln1, uint16_t, M * D); TALLOC(s.att, uint16_t, M * D); TALLOC(s.ln2, uint16_t, M * D);
        TALLOC(s.q, uint16_t, B * H * npad_ * 64); TALLOC(s.k, uint16_t, B * H * npad_ * 64); TALLOC(s.vt, uint16_t, B * H * 64 * npad_);
        TALLOC(s.pre, uint16_t, M * 4 * D); TALLOC(s.mlp, uint16_t, M * 4 * D);
    }
    TALLOC(xlast_, float, M * D);
    size_t rows_max = B * 4 * lh_[0] * lw_[0] * F;               // largest row-major 16-bit temporary (d path_1 / d up_1)
    size_t wsa = 0, wsb = 0, wdw = 0;
    auto lin_need = [&](size_t m, size_t n, size_t k) { wsa = std::max(wsa, n * up64(m)); wsb = std::max(wsb, k * up64(m)); };
    auto conv_need = [&](size_t b, size_t h, size_t w, size_t ci, size_t co) {
        const size_t mpp = up64(b * (h + 2) * (w + 2));
        wsa = std::max(wsa, co * mpp); wsb = std::max(wsb, 9 * ci * mpp); wdw = std::max(wdw, co * 9 * ci);
    };
    lin_need(M, 4 * D, 4 * D); lin_need(M, 3 * D, D); lin_need(Mr, D, 2 * D); lin_need(Mr, D, 3 * c.patch * c.patch);
    lin_need(B * 4 * lh_[0] * lw_[0], c.out_c, F);
    for (int l = 0; l < 4; ++l) {
        LevelSave& v = lv_[l];
        const size_t Cp = cp_[l];
        TALLOC(v.cat, uint16_t, Mr * 2 * D); TALLOC(v.ropre, uint16_t, Mr * D); TALLOC(v.ro, uint16_t, Mr * D); TALLOC(v.dro, uint16_t, Mr * D);
        TALLOC(v.r1, uint16_t, Mr * Cp);
        if (c.resample_kind[l] == LSEG_RS_CONV_S2) TALLOC(v.tmp, uint16_t, B * (gh_ + 2) * (gw_ + 2) * Cp);
        const size_t pp = B * (lh_[l] + 2) * (lw_[l] + 2);
        TALLOC(drn_[l], uint16_t, pp * F); TALLOC(dL_[l], uint16_t, pp * Cp);
        TALLOC(dmapA_[l], uint16_t, pp * F); TALLOC(dmapB_[l], uint16_t, pp * F); TALLOC(dmapC_[l], uint16_t, pp * F); TALLOC(dmapD_[l], uint16_t, pp * F);
        for (int u = 0; u < 2; ++u) {
            Rcu& U = u == 0 ? refine_[l].u1 : refine_[l].u2;
            if (u == 0 && l == 3) continue;
            TALLOC(U.cv1, uint16_t, pp * F); TALLOC(U.n1, uint16_t, pp * F); TALLOC(U.cv2, uint16_t, pp * F);
            TALLOC(U.st1, float, 2 * F); TALLOC(U.st2, float, 2 * F);
        }
        conv_need(B, lh_[l], lw_[l], F, F);
        conv_need(B, lh_[l], lw_[l], Cp, F);                     // layerN_rn
        lin_need(B * 4 * lh_[l] * lw_[l], F, F);                 // out_conv
        lin_need(Mr, Cp, D);                                     // 1x1 reassemble conv
        if (c.resample_kind[l] == LSEG_RS_CONVT) {
            const size_t s2 = (size_t)c.resample_k[l] * c.resample_k[l];
            lin_need(Mr, s2 * Cp, Cp);
            wdw = std::max(wdw, s2 * Cp * Cp);
            rows_max = std::max(rows_max, Mr * s2 * Cp);
        } else if (c.resample_kind[l] == LSEG_RS_CONV_S2) {
            conv_need(B, gh_, gw_, Cp, Cp);
            TALLOC(ddil_, uint16_t, B * (gh_ + 2) * (gw_ + 2) * Cp);
            TALLOC(dtmp_, uint16_t, B * (gh_ + 2) * (gw_ + 2) * Cp);
        }
        wdw = std::max(wdw, Cp * D);
    }
    rows_max = std::max(rows_max, Mr * 2 * D);
    TALLOC(rowsA_, uint16_t, rows_max); TALLOC(rowsB_, uint16_t, rows_max);
    TALLOC(dpath0_, uint16_t, B * 4 * lh_[0] * lw_[0] * F);
    ws_a_n_ = wsa; ws_b_n_ = wsb; ws_dw_n_ = wdw;
    TALLOC(ws_a_, uint16_t, wsa); TALLOC(ws_b_, uint16_t, wsb); TALLOC(ws_dw_, float, wdw);
    ws_part_n_ = (size_t)16 << 20;                   // 64 MB of fp32 split-K partials
    TALLOC(ws_part_, float, ws_part_n_);
    TALLOC(ws_stats_, float, 2 * std::max<size_t>(F, 16 * 1024));
    TALLOC(zeros_, float, 16 * 1024);
    TALLOC(ws_ln_, float, (size_t)LN_BWD_PARTIAL_BLOCKS * 2 * D);
    if (c.flags & 8) {                               // deterministic reductions: partial rows of the column-statistics kernels (bias
        ws_det_n_ = (size_t)1 << 20;                 // gradients, BatchNorm batch sums forward / backward) instead of fp32 atomics
        TALLOC(ws_det_, float, ws_det_n_);
    }
    TALLOC(gx_, float, M * D); TALLOC(dpos_, float, (size_t)ntok_ * D);
    TALLOC(attn_ws_, char, attention_backward_ws_bytes((int)B, (int)H, npad_));
    TALLOC(g16_, uint16_t, M * D); TALLOC(dmlp_, uint16_t, M * 4 * D); TALLOC(dln_, uint16_t, M * D); TALLOC(datt_, uint16_t, M * D);
    TALLOC(dqkv_, uint16_t, M * 3 * D); TALLOC(dtok_, uint16_t, Mr * D);
    const size_t hw1 = (size_t)4 * lh_[0] * lw_[0], Kp = up64(c.max_labels);
    TALLOC(lse_px_, float, B * 4 * hw1);
    TALLOC(drows_, uint16_t, B * hw1 * Kp); TALLOC(da_, uint16_t, B * hw1 * c.out_c); TALLOC(df_, uint16_t, B * hw1 * c.out_c);
    TALLOC(tnT_, uint16_t, (size_t)c.out_c * Kp); TALLOC(tn16_, uint16_t, (size_t)c.max_labels * c.out_c);
    TALLOC(counts_, unsigned long long, 2 + 3 * (size_t)c.max_labels); TALLOC(nll_, double, 2);
    LSEG_HIP_TRY(hipDeviceSynchronize());
    train_alloc_ = true;
    return 0;
}

// ---- train-mode parameter packs: transposed Linear weights (dgrad), raw + flipped 3x3 weights, BatchNorm affine ----------------
int Engine::finalize_train(hipStream_t st) {
    const lseg_config& c = cfg;
    const int F = c.features;
    std::vector<Lin*> wts;                       // every Linear whose dgrad needs wt [k, n] = w^T
    auto make_wd = [&](Lin& L, int co, int ci) -> int {   // w [co, 9, ci] -> wd [ci, 9, co] flipped
        if (!L.wd) TALLOC(L.wd, uint16_t, (size_t)co * 9 * ci);
        return launch_conv_dgrad_pack(L.w, L.wd, co, ci, st);
    };
    for (auto& b : blocks_) { wts.push_back(&b.qkv); wts.push_back(&b.proj); wts.push_back(&b.fc1); wts.push_back(&b.fc2); }
    for (int l = 0; l < 4; ++l) {
        wts.push_back(&readout_[l]); wts.push_back(&r1x1_[l]);
        if (c.resample_kind[l] == LSEG_RS_CONVT) wts.push_back(&rsmp_[l]);
        else if (c.resample_kind[l] == LSEG_RS_CONV_S2) TRY(make_wd(rsmp_[l], cp_[l], cp_[l]));
        TRY(make_wd(layer_rn_[l], F, cp_[l]));
        Refine& R = refine_[l];
        wts.push_back(&R.out_conv);
        for (int u = 0; u < 2; ++u) {
            if (u == 0 && !R.has_u1) continue;
            Rcu& U = u == 0 ? R.u1 : R.u2;
            char buf[96];
            snprintf(buf, sizeof(buf), "scratch.refinenet%d.resConfUnit%d.", l + 1, u + 1);
            U.key = buf;
            TRY(pack_conv3(U.key + "conv1.weight", "", "", F, F, F, F, U.r1, st));
            TRY(pack_conv3(U.key + "conv2.weight", "", "", F, F, F, F, U.r2, st));
            TRY(make_wd(U.r1, F, F)); TRY(make_wd(U.r2, F, F));
            TRY(pack_f32(U.key + "bn1.weight", F, U.g1, st)); TRY(pack_f32(U.key + "bn1.bias", F, U.be1, st));
            TRY(pack_f32(U.key + "bn2.weight", F, U.g2, st)); TRY(pack_f32(U.key + "bn2.bias", F, U.be2, st));
        }
    }
    wts.push_back(&head1_);
    // the transposes: one table-driven launch for every matrix with 16-byte rows on both sides, single launches for the rest
    if (!wt_table_) {
        std::vector<TransposeJob> jobs;
        unsigned blk = 0;
        for (Lin* L : wts) {
            if (!L->wt) TALLOC(L->wt, uint16_t, (size_t)L->n * L->k);
            if ((L->n & 7) || (L->k & 7)) continue;
            TransposeJob j;
            j.src = L->w; j.dst = L->wt; j.R = L->n; j.C = L->k; j.blk0 = blk; j.tiles_r = (L->n + 63) / 64;
            blk += (unsigned)j.tiles_r * (unsigned)((L->k + 63) / 64);
            jobs.push_back(j);
        }
        wt_n_ = (int)jobs.size(); wt_blocks_ = blk;
        if (wt_n_) {
            wt_table_ = (TransposeJob*)dalloc(jobs.size() * sizeof(TransposeJob));
            if (!wt_table_) return set_error(LSEG_ERR_HIP, "out of device memory (transpose table)");
            LSEG_HIP_TRY(hipMemcpy(wt_table_, jobs.data(), jobs.size() * sizeof(TransposeJob), hipMemcpyHostToDevice));
        }
    }
    TRY(launch_transpose16_multi(wt_table_, wt_n_, wt_blocks_, st));
    for (Lin* L : wts)
        if ((L->n & 7) || (L->k & 7)) TRY(launch_transpose16(L->w, L->wt, L->n, L->k, L->k, L->n, st));
    if (!partial_pack_) LSEG_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// ---- ResidualConvUnit_custom in train() mode (lseg_blocks.py:265-288): out = bn2(conv2(relu(bn1(conv1(relu(x)))))) + x [+ res2] ----
// ResidualConvUnit in train mode.  The 3x3 convs read ReLU(x): clamping the operand fragments inside the K-loop costs the forward conv a
// third of its MFMA rate and the K-major weight-gradient GEMM more (15 % MFMA busy against 24 %, profiles/r03_train_pmc_summary.txt), so
// the ReLU-ed maps are materialised by whoever produces them: `in_relu` = ReLU(in) from the producing conv's / BatchNorm's epilogue (NULL:
// clamp in the K-loop), U.n1 holds ReLU(bn1(conv1)) only -- its sign pattern is all the backward needs --, `out_relu` (optional) receives
// ReLU(out) for a following unit.
int Engine::rcu_train(const uint16_t* in, const uint16_t* in_relu, Rcu& U, const uint16_t* res2, uint16_t* out, uint16_t* out_relu, int B, int H,
                      int W, hipStream_t st) {
    const int F = cfg.features;
    const double cnt = (double)B * H * W * bn_world;
    BoundParam rm, rv;
    Lin c1 = U.r1; c1.b = zeros_;
    Lin c2 = U.r2; c2.b = zeros_;
    U.in_relu = in_relu;
    if (in_relu) TRY(conv3x3(in_relu, c1, nullptr, nullptr, U.cv1, B, H, W, 1, 0, 0, st));
    else TRY(conv3x3(in, c1, nullptr, nullptr, U.cv1, B, H, W, 1, 1, 0, st));
    TRY(launch_bn_stats(U.cv1, U.st1, B, H, W, F, img_dt_, st, ws_det_ ? 1 : zero_note(zero_fwd_, U.st1, (size_t)2 * F) ? 1 : 0, ws_det_, ws_det_n_));
    TRY(bn_sync(U.st1, 2 * F, st));
    TRY(launch_bn_apply(U.cv1, nullptr, U.st1, U.g1, U.be1, nullptr, nullptr, B, H, W, F, 1e-5f, cnt, img_dt_, st, U.n1));
    TRY(conv3x3(U.n1, c2, nullptr, nullptr, U.cv2, B, H, W, 1, 0, 0, st));
    TRY(launch_bn_stats(U.cv2, U.st2, B, H, W, F, img_dt_, st, ws_det_ ? 1 : zero_note(zero_fwd_, U.st2, (size_t)2 * F) ? 1 : 0, ws_det_, ws_det_n_));
    TRY(bn_sync(U.st2, 2 * F, st));
    TRY(launch_bn_apply(U.cv2, out, U.st2, U.g2, U.be2, in, res2, B, H, W, F, 1e-5f, cnt, img_dt_, st, out_relu));
    // running statistics live in the caller's tensors (momentum 0.1, unbiased variance: nn.BatchNorm2d / SyncBatchNorm)
    for (int k = 1; k <= 2; ++k) {
        auto m = bound_.find(U.key + "bn" + std::to_string(k) + ".running_mean"), v = bound_.find(U.key + "bn" + std::to_string(k) + ".running_var");
        if (m == bound_.end() || v == bound_.end() || m->second.dtype != LSEG_F32 || v->second.dtype != LSEG_F32) continue;
        TRY(launch_bn_running_update(k == 1 ? U.st1 : U.st2, (float*)m->second.ptr, (float*)v->second.ptr, F, cnt, 0.1f, st));
    }
    return 0;
}

int Engine::forward_train(const float* x_in, int B, float* logits, hipStream_t st) {
    const lseg_config& c = cfg;
    const int D = c.dim, H = c.heads, F = c.features, M = B * ntok_;
    if (group_k > 0) return set_error(LSEG_ERR_UNSUPPORTED, "train mode with per-image label sets is not implemented");
    if (!train_alloc_) return set_error(LSEG_ERR_STATE, "train mode was not enabled (lseg_set_train)");
    train_fwd_valid_ = false;
    loss_target_ = nullptr;
    last_B_ = B;
    low_pending_ = false;
    eval_stale_ = true;             // the BatchNorm running statistics move: the eval-mode (BN-folded) packs are refreshed by the next eval forward
    const bool run_text = !text_external_ && (!text_cache || !text_valid);
    if (run_text) {
        LSEG_HIP_TRY(hipEventRecord(ev_fork_, st));
        LSEG_HIP_TRY(hipStreamWaitEvent(text_stream_, ev_fork_, 0));
        TRY(encode_text(text_stream_));
        LSEG_HIP_TRY(hipEventRecord(ev_join_, text_stream_));
    }
    TRY(zero_begin(zero_fwd_, st));
    TRY(launch_im2col_patch(x_in, patchA_, B, c.img_h, c.img_w, c.patch, img_dt_, st));
    GemmArgs g;
    gemm_args_init(g);
    g.A = patchA_; g.W = patch_.w; g.M = B * np_; g.N = D; g.K = patch_.k; g.lda = patch_.k; g.ldw = patch_.k;
    g.bias = patch_.b; g.res_mode = RES_PERIODIC; g.res = pos_; g.res_dtype = DT_F32; g.ldr = D;
    g.C = sv_[0].xin; g.out_dtype = DT_F32; g.ldc = D; g.map_mode = MAP_PERIODIC; g.p_div = np_; g.p_mul = ntok_; g.p_off = 1;
    TRY(launch_gemm(g, img_dt_, st));
    TRY(launch_cls_rows(cls_, pos_, sv_[0].xin, B, ntok_, D, st));
    for (int i = 0; i < c.depth; ++i) {
        VitBlock& b = blocks_[i];
        BlockSave& s = sv_[i];
        float* xout = i + 1 < c.depth ? sv_[i + 1].xin : xlast_;
        TRY(launch_layernorm(s.xin, DT_F32, b.g1, b.b1, s.ln1, img_dt_, M, D, 1e-6f, st));
        gemm_args_init(g);
        g.A = s.ln1; g.W = b.qkv.w; g.M = M; g.N = 3 * D; g.K = D; g.lda = D; g.ldw = D;
        g.bias = b.qkv.b; g.out_dtype = img_dt_; g.map_mode = MAP_QKV;
        g.C = s.q; g.Ck = s.k; g.Cv = s.vt; g.qkv_dim = D; g.qkv_ntok = ntok_; g.qkv_npad = npad_; g.qkv_heads = H;
        TRY(launch_gemm(g, img_dt_, st));
        TRY(launch_attention_lse(s.q, s.k, s.vt, s.att, s.lse, B, H, ntok_, npad_, img_dt_, 0, 0.125f, st));
        gemm_args_init(g);
        g.A = s.att; g.W = b.proj.w; g.M = M; g.N = D; g.K = D; g.lda = D; g.ldw = D;
        g.bias = b.proj.b; g.res_mode = RES_DEST; g.res = s.xin; g.res_dtype = DT_F32;
        g.C = s.xmid; g.out_dtype = DT_F32; g.ldc = D; g.map_mode = MAP_LINEAR;
        TRY(launch_gemm(g, img_dt_, st));
        TRY(launch_layernorm(s.xmid, DT_F32, b.g2, b.b2, s.ln2, img_dt_, M, D, 1e-6f, st));
        gemm_args_init(g);
        g.A = s.ln2; g.W = b.fc1.w; g.M = M; g.N = 4 * D; g.K = D; g.lda = D; g.ldw = D;
        g.bias = b.fc1.b; g.out_dtype = img_dt_; g.ldc = 4 * D; g.map_mode = MAP_LINEAR;
        TRY(linear_gelu_saved(g, s.pre, s.mlp, st));
        gemm_args_init(g);
        g.A = s.mlp; g.W = b.fc2.w; g.M = M; g.N = D; g.K = 4 * D; g.lda = 4 * D; g.ldw = 4 * D;
        g.bias = b.fc2.b; g.res_mode = RES_DEST; g.res = s.xmid; g.res_dtype = DT_F32;
        g.C = xout; g.out_dtype = DT_F32; g.ldc = D; g.map_mode = MAP_LINEAR;
        TRY(launch_gemm(g, img_dt_, st));
        for (int l = 0; l < 4; ++l) {
            if (c.hooks[l] != i) continue;
            LevelSave& v = lv_[l];
            const int C = cp_[l];
            TRY(launch_readout_cat(xout, v.cat, B, ntok_, D, img_dt_, st));
            gemm_args_init(g);
            g.A = v.cat; g.W = readout_[l].w; g.M = B * np_; g.N = D; g.K = 2 * D; g.lda = 2 * D; g.ldw = 2 * D;
            g.bias = readout_[l].b; g.out_dtype = img_dt_; g.ldc = D; g.map_mode = MAP_LINEAR;
            TRY(linear_gelu_saved(g, v.ropre, v.ro, st));
            gemm_args_init(g);
            g.A = v.ro; g.W = r1x1_[l].w; g.M = B * np_; g.N = C; g.K = D; g.lda = D; g.ldw = D;
            g.bias = r1x1_[l].b; g.out_dtype = img_dt_; g.ldc = C;
            if (c.resample_kind[l] == LSEG_RS_CONVT) { g.C = v.r1; g.map_mode = MAP_LINEAR; }
            else if (c.resample_kind[l] == LSEG_RS_IDENTITY) { g.C = L_[l]; g.map_mode = MAP_PADDED; g.ho = gh_; g.wo = gw_; }
            else { g.C = v.tmp; g.map_mode = MAP_PADDED; g.ho = gh_; g.wo = gw_; }
            TRY(launch_gemm(g, img_dt_, st));
            if (c.resample_kind[l] == LSEG_RS_CONVT) {
                const int s2 = c.resample_k[l];
                gemm_args_init(g);
                g.A = v.r1; g.W = rsmp_[l].w; g.M = B * np_; g.N = s2 * s2 * C; g.K = C; g.lda = C; g.ldw = C;
                g.bias = rsmp_[l].b; g.bias_mod = C; g.C = L_[l]; g.out_dtype = img_dt_; g.ldc = C;
                g.map_mode = MAP_PIXSHUF; g.ho = gh_; g.wo = gw_; g.ps_s = s2; g.ps_C = C;
                TRY(launch_gemm(g, img_dt_, st));
            } else if (c.resample_kind[l] == LSEG_RS_CONV_S2) {
                TRY(conv3x3(v.tmp, rsmp_[l], nullptr, nullptr, L_[l], B, gh_, gw_, 2, 0, 0, st));
            }
            TRY(conv3x3(L_[l], layer_rn_[l], nullptr, nullptr, rn_[l], B, lh_[l], lw_[l], 1, 0, 0, st, rnr_[l], &rn_relu_ok_[l]));
        }
    }
    // refinenet4..1 in train mode
    for (int r = 4; r >= 1; --r) {
        const int l = r - 1, Hh = lh_[l], Ww = lw_[l];
        Refine& R = refine_[l];
        const uint16_t *in2 = rn_[l], *in2r = rn_relu_ok_[l] ? rnr_[l] : nullptr;
        if (R.has_u1) { TRY(rcu_train(rn_[l], in2r, R.u1, path_[l + 1], sum_[l], sumr_[l], B, Hh, Ww, st)); in2 = sum_[l]; in2r = sumr_[l]; }
        TRY(rcu_train(in2, in2r, R.u2, nullptr, t2_[l], nullptr, B, Hh, Ww, st));
        TRY(launch_upsample2x_nhwc(t2_[l], up_[l], B, Hh, Ww, F, img_dt_, st));
        gemm_args_init(g);
        g.A = up_[l]; g.W = R.out_conv.w; g.M = B * 4 * Hh * Ww; g.N = F; g.K = F; g.lda = F; g.ldw = F;
        g.bias = R.out_conv.b; g.C = path_[l]; g.out_dtype = img_dt_; g.ldc = F;
        if (l > 0) { g.map_mode = MAP_PADDED; g.ho = 2 * Hh; g.wo = 2 * Ww; }
        else g.map_mode = MAP_LINEAR;
        TRY(launch_gemm(g, img_dt_, st));
    }
    if (run_text) LSEG_HIP_TRY(hipStreamWaitEvent(st, ev_join_, 0));
    const int h1 = 2 * lh_[0], w1 = 2 * lw_[0], hw1 = h1 * w1, Mp = B * hw1;
    const float logit_scale = expf(logf(1.0f / 0.07f));
    gemm_args_init(g);
    g.A = path_[0]; g.W = head1_.w; g.M = Mp; g.N = c.out_c; g.K = F; g.lda = F; g.ldw = F;
    g.bias = head1_.b; g.C = feat_; g.out_dtype = DT_F32; g.ldc = c.out_c; g.map_mode = MAP_LINEAR;
    TRY(launch_gemm(g, img_dt_, st));
    TRY(launch_l2norm_scale_f16(feat_, a16_, Mp, c.out_c, logit_scale, st));
    gemm_args_init(g);
    g.A = tnorm_; g.W = a16_; g.M = K_; g.N = Mp; g.K = c.out_c; g.lda = c.out_c; g.ldw = c.out_c;
    g.round_mid = 1; g.C = low_; g.out_dtype = DT_F32; g.map_mode = MAP_LABELPLANES; g.p_div = hw1;
    TRY(launch_gemm(g, DT_F16, st));
    // the text features as the dgrad operand of the correlation: bf16, transposed, K padded to the GEMM's K-step
    const int Kp = (int)up64(K_);
    if (cfg.flags & 2) {
        TRY(launch_convert(tnorm_, DT_F16, tn16_, DT_BF16, (size_t)K_ * c.out_c, st));
        TRY(launch_transpose16(tn16_, tnT_, K_, c.out_c, c.out_c, Kp, st));
    } else {
        TRY(launch_transpose16(tnorm_, tnT_, K_, c.out_c, c.out_c, Kp, st));      // fp16 as it is: the reference's dgrad is a half x half product
    }
    // output_conv (x2 bilinear) only when the caller wants the logits: the loss and its gradient are taken on the low-resolution ones
    if (logits) TRY(launch_upsample2x_planes(low_, logits, B * K_, h1, w1, st));
    last_low_ = low_; last_kout_ = K_;
    train_B_ = B;
    train_fwd_valid_ = true;
    TRY(zero_end(zero_fwd_));
    return 0;
}

// split-K plan of a weight-gradient GEMM [M x N] over nk K-steps: enough (tile, K-range) work items to fill the chip twice with
// 128x128 tiles, ranges of at least 8 K-steps, partials within the workspace.  Fills g.nsplit / g.split_steps; returns the split count.
int Engine::pick_split(int M, int N, int nk, GemmArgs& g) {
    static const bool off = getenv("LSEG_NO_SPLITK") != nullptr;        // tools: bisecting switch
    if (off) return 1;
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    long ns = (2L * 256 + tiles - 1) / tiles;
    if (ns > nk / 8) ns = nk / 8;
    while (ns > 1 && (size_t)ns * M * N > ws_part_n_) --ns;
    if (ns <= 1) return 1;
    const int steps = (int)((nk + ns - 1) / ns);
    ns = (nk + steps - 1) / steps;                  // every range non-empty, the last one possibly shorter
    if (ns <= 1) return 1;
    g.nsplit = (int)ns; g.split_steps = steps;
    return (int)ns;
}

// ---- weight gradient straight from the row-major operands (gemm.hip "K-MAJOR operands") ------------------------------------------------
// split-K plan of a K-major wgrad [rows x cols] over nk K-steps: 128 x 128 tiles cut into ~2 work items per CU of >= 8 K-steps.
// (256 x 256 K-major tiles were built, parity-tested and measured 2 % slower on the whole training step in round 3 --
// profiles/r03_train_experiments.txt -- and removed in round 4.)
static void plan_kmajor(GemmArgs& g, int rows, int cols, int nk, size_t ws_floats) {
    const long tiles = (long)((rows + 127) / 128) * (cols / 128);
    long ns = (512 + tiles - 1) / tiles;
    if (ns > nk / 8) ns = nk / 8;
    while (ns > 1 && (size_t)ns * rows * cols > ws_floats) --ns;
    if (ns < 1) ns = 1;
    const int steps = (int)((nk + ns - 1) / ns);
    g.tile_hint = 2;
    g.nsplit = (int)((nk + steps - 1) / steps);
    g.split_steps = steps;
    g.c_split_stride = (size_t)rows * cols;
}

bool wgrad_kmajor_ok(int rows_out, int cols_out, size_t ws_floats) {
    return rows_out >= 1 && (cols_out % 128) == 0 && (size_t)rows_out * cols_out <= ws_floats;
}
int launch_wgrad_kmajor(const void* dy, int ldy, const void* x, int ldx, int M, int rows_out, int cols_out, float* dw, int accumulate,
                        float* ws, size_t ws_floats, int ab_dtype, hipStream_t st) {
    if (!wgrad_kmajor_ok(rows_out, cols_out, ws_floats)) return set_error(LSEG_ERR_UNSUPPORTED, "wgrad_kmajor: %d x %d", rows_out, cols_out);
    GemmArgs g;
    gemm_args_init(g);
    const int Kp = (M + 63) / 64 * 64, nk = Kp / 64;
    g.A = (const uint16_t*)dy; g.lda = ldy; g.W = (const uint16_t*)x; g.ldw = ldx;
    g.M = rows_out; g.N = cols_out; g.K = Kp; g.kmajor = 1; g.k_valid = M;
    g.C = ws; g.out_dtype = DT_F32; g.ldc = cols_out; g.map_mode = MAP_LINEAR;
    plan_kmajor(g, rows_out, cols_out, nk, ws_floats);
    const int ns = g.nsplit;
    TRY(launch_gemm(g, ab_dtype, st));
    return launch_sum_partials(ws, dw, ns, (size_t)rows_out * cols_out, g.c_split_stride, accumulate, st);
}

int launch_conv_wgrad_kmajor(const void* dy_pad, const void* x_pad, int relu_x, int B, int H, int W, int Cin, int Cout, float* ws, size_t ws_floats,
                             int ab_dtype, int* ns_out, hipStream_t st) {
    if ((Cin % 128) != 0 || !wgrad_kmajor_ok(Cout, 9 * Cin, ws_floats)) return set_error(LSEG_ERR_UNSUPPORTED, "conv_wgrad_kmajor: Cin=%d Cout=%d", Cin, Cout);
    const int Mp = B * (H + 2) * (W + 2), nk = (Mp + 63) / 64;
    GemmArgs g;
    gemm_args_init(g);
    g.A = (const uint16_t*)dy_pad; g.lda = Cout; g.W = (const uint16_t*)x_pad; g.ldw = Cin; g.M = Cout; g.N = 9 * Cin; g.K = nk * 64;
    g.kmajor = 1; g.k_valid = Mp; g.kconv_cin = Cin; g.kconv_wp = W + 2; g.relu_in = relu_x ? 1 : 0;
    g.C = ws; g.out_dtype = DT_F32; g.ldc = 9 * Cin; g.map_mode = MAP_LINEAR;
    plan_kmajor(g, Cout, 9 * Cin, nk, ws_floats);
    const int ns = g.nsplit;
    if (ns_out) *ns_out = ns;
    return launch_gemm(g, ab_dtype, st);
}

// Linear + GELU of the training forward: z = x W^T + b is kept for the backward, gelu(z) feeds the next Linear.  One launch where the
// GEMM's specialised epilogue fits (both stores from the accumulators), else the GEMM and a GELU pass over its output.
int Engine::linear_gelu_saved(GemmArgs& g, uint16_t* pre, uint16_t* out, hipStream_t st) {
    g.act = ACT_GELU; g.C = out; g.C_pre = pre;
    if (gemm_fuses_gelu(g, img_dt_)) return launch_gemm(g, img_dt_, st);
    g.act = ACT_NONE; g.C = pre; g.C_pre = nullptr;
    TRY(launch_gemm(g, img_dt_, st));
    return launch_gelu_forward(pre, out, (size_t)g.M * g.N, img_dt_, st);
}

// ---- Linear backward on the forward MFMA kernel (contraction dimension transposed onto the fast axis) ---------------------------
//   dx [M,K] = dy [M,N] . W [N,K]     (A = dy, "weights" = wt = W^T [K,N])
//   dw [N,K] = dy^T . x               (A = dy^T [N,Mp], "weights" = x^T [K,Mp]; fp32, written in the parameter's own layout)
//   db [N]   = column sums of dy
int Engine::lin_bwd(const uint16_t* dy, int M, int N, int K, const uint16_t* x, const uint16_t* wt, uint16_t* dx, float* dw, float* db,
                    int acc, hipStream_t st, int dw_rows, const uint16_t* dgelu_pre) {
    GemmArgs g;
    if (dx) {
        gemm_args_init(g);
        g.A = dy; g.W = wt; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldw = N;
        g.bias = zeros_; g.C = dx; g.out_dtype = img_dt_; g.ldc = K; g.map_mode = MAP_LINEAR;
        // dgelu_pre: x = gelu(z) came out of a GELU; dx leaves as d z = (dy . W) * gelu'(z), in the GEMM epilogue where it can
        g.dgelu_pre = dgelu_pre;
        const bool fused = dgelu_pre && gemm_fuses_gelu(g, img_dt_);
        if (!fused) g.dgelu_pre = nullptr;
        TRY(launch_gemm(g, img_dt_, st));
        if (dgelu_pre && !fused) TRY(launch_gelu_backward(dx, dgelu_pre, dx, (size_t)M * K, img_dt_, st));
    }
    if (dw && wgrad_kmajor_ok(dw_rows > 0 ? dw_rows : N, K, ws_part_n_)) {
        TRY(launch_wgrad_kmajor(dy, N, x, K, M, dw_rows > 0 ? dw_rows : N, K, dw, acc, ws_part_, ws_part_n_, img_dt_, st));
    } else if (dw) {
        const int Mp = (int)up64(M);
        if ((size_t)N * Mp > ws_a_n_ || (size_t)K * Mp > ws_b_n_) return set_error(LSEG_ERR_STATE, "wgrad workspace too small (%d x %d x %d)", M, N, K);
        TRY(launch_transpose16(dy, ws_a_, M, N, N, Mp, st));
        TRY(launch_transpose16(x, ws_b_, M, K, K, Mp, st));
        gemm_args_init(g);
        g.A = ws_a_; g.W = ws_b_; g.M = dw_rows > 0 ? dw_rows : N; g.N = K; g.K = Mp; g.lda = Mp; g.ldw = Mp;
        g.C = dw; g.out_dtype = DT_F32; g.ldc = K; g.map_mode = MAP_LINEAR;
        const int ns = pick_split(g.M, g.N, Mp / 64, g);
        if (ns > 1) {      // small output, long contraction: split-K partials, summed (and accumulated) afterwards
            g.C = ws_part_; g.c_split_stride = (size_t)g.M * K;
            TRY(launch_gemm(g, img_dt_, st));
            TRY(launch_sum_partials(ws_part_, dw, ns, (size_t)g.M * K, g.c_split_stride, acc, st));
        } else {
            if (acc) { g.res_mode = RES_DEST; g.res = dw; g.res_dtype = DT_F32; }
            TRY(launch_gemm(g, img_dt_, st));
        }
    }
    if (db) TRY(bias_sum(dy, db, M, dw_rows > 0 ? dw_rows : N, N, acc, st));
    return 0;
}

// ---- 3x3 stride-1 conv backward in the padded-NHWC layout: dx = conv(dy, flipped weights); dw = dy^T x (9 shifted transposes of x) ----
int Engine::conv_bwd(const uint16_t* dy_pad, const uint16_t* x_pad, int relu_x, const Lin& w, uint16_t* dx_pad, float* dw_dst, int B, int H,
                     int W, int Cin, int Cout, int Ci_real, int Co_real, int acc, hipStream_t st) {
    if (dx_pad) {
        Lin d; d.w = w.wd; d.b = zeros_; d.n = Cin; d.k = 9 * Cout;
        TRY(conv3x3(dy_pad, d, nullptr, nullptr, dx_pad, B, H, W, 1, 0, 0, st));
    }
    if (dw_dst && (Cin % 128) == 0 && wgrad_kmajor_ok(Cout, 9 * Cin, ws_part_n_)) {
        // dW[co][tap][ci] = sum_m dY[m][co] * relu?(X)[m + shift(tap)][ci] straight from the padded NHWC maps (K-major GEMM operands)
        int ns = 1;
        TRY(launch_conv_wgrad_kmajor(dy_pad, x_pad, relu_x, B, H, W, Cin, Cout, ws_part_, ws_part_n_, img_dt_, &ns, st));
        TRY(launch_conv_wgrad_unpack(ws_part_, dw_dst, Co_real, Ci_real, Cin, acc, st, ns, (size_t)Cout * 9 * Cin));
    } else if (dw_dst) {
        const int Mp = B * (H + 2) * (W + 2), Mpp = (int)up64(Mp);
        if ((size_t)Cout * Mpp > ws_a_n_ || (size_t)9 * Cin * Mpp > ws_b_n_ || (size_t)Cout * 9 * Cin > ws_dw_n_)
            return set_error(LSEG_ERR_STATE, "conv wgrad workspace too small");
        TRY(launch_transpose16(dy_pad, ws_a_, Mp, Cout, Cout, Mpp, st));
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * (W + 2) + (t % 3 - 1);
            TRY(launch_transpose16(x_pad, ws_b_ + (size_t)t * Cin * Mpp, Mp, Cin, Cin, Mpp, st, shift, relu_x));
        }
        GemmArgs g;
        gemm_args_init(g);
        g.A = ws_a_; g.W = ws_b_; g.M = Cout; g.N = 9 * Cin; g.K = Mpp; g.lda = Mpp; g.ldw = Mpp;
        g.C = ws_dw_; g.out_dtype = DT_F32; g.ldc = 9 * Cin; g.map_mode = MAP_LINEAR;
        const int ns = pick_split(Cout, 9 * Cin, Mpp / 64, g);
        if (ns > 1) { g.C = ws_part_; g.c_split_stride = (size_t)Cout * 9 * Cin; }
        TRY(launch_gemm(g, img_dt_, st));
        TRY(launch_conv_wgrad_unpack(ns > 1 ? ws_part_ : ws_dw_, dw_dst, Co_real, Ci_real, Cin, acc, st, ns, g.c_split_stride));
    }
    return 0;
}

// backward of rcu_train: dout -> din (= relu'(in) o d_a0 + dout), parameter gradients of conv1/bn1/conv2/bn2
int Engine::rcu_backward(const uint16_t* dout, const uint16_t* in, Rcu& U, uint16_t* din, int lev, int B, int H, int W, int acc, hipStream_t st) {
    const int F = cfg.features;
    const double cnt = (double)B * H * W * bn_world;
    const size_t nmap = (size_t)B * (H + 2) * (W + 2) * F;
    uint16_t* dC = dmapC_[lev];
    uint16_t* dD = dmapD_[lev];
    float* bst = ws_stats_;
    // bn2
    TRY(launch_bn_bwd_stats(dout, U.cv2, U.st2, bst, B, H, W, F, 1e-5f, cnt, img_dt_, st, ws_det_, ws_det_n_));
    TRY(launch_fold_rows(bst, grad(U.key + "bn2.bias", F), 1, F, F, acc, st));
    TRY(launch_fold_rows(bst + F, grad(U.key + "bn2.weight", F), 1, F, F, acc, st));
    TRY(bn_sync(bst, 2 * F, st));
    TRY(launch_bn_bwd_apply(dout, U.cv2, U.st2, bst, U.g2, dC, B, H, W, F, 1e-5f, cnt, img_dt_, st));
    // conv2 (input relu(n1): U.n1 holds the ReLU-ed map, whose sign pattern is n1's)
    TRY(conv_bwd(dC, U.n1, 0, U.r2, dD, grad(U.key + "conv2.weight", (size_t)F * F * 9), B, H, W, F, F, F, F, acc, st));
    TRY(launch_relu_backward(dD, U.n1, dD, nmap, st));
    // bn1
    TRY(launch_bn_bwd_stats(dD, U.cv1, U.st1, bst, B, H, W, F, 1e-5f, cnt, img_dt_, st, ws_det_, ws_det_n_));
    TRY(launch_fold_rows(bst, grad(U.key + "bn1.bias", F), 1, F, F, acc, st));
    TRY(launch_fold_rows(bst + F, grad(U.key + "bn1.weight", F), 1, F, F, acc, st));
    TRY(bn_sync(bst, 2 * F, st));
    TRY(launch_bn_bwd_apply(dD, U.cv1, U.st1, bst, U.g1, dC, B, H, W, F, 1e-5f, cnt, img_dt_, st));
    // conv1 (input relu(in)); din = relu'(in) o d_a0 + dout
    TRY(conv_bwd(dC, U.in_relu ? U.in_relu : in, U.in_relu ? 0 : 1, U.r1, dD, grad(U.key + "conv1.weight", (size_t)F * F * 9), B, H, W, F, F, F, F, acc, st));
    TRY(launch_relu_backward_add(dD, in, dout, din, nmap, img_dt_, st));
    return 0;
}

// FeatureFusionBlock_custom backward (refinenet r): consumes d path_r, produces d layer_r_rn -> d L_r and d path_{r+1}
int Engine::refine_backward(int r, int B, int acc, hipStream_t st) {
    const int l = r - 1, H = lh_[l], W = lw_[l], F = cfg.features;
    Refine& R = refine_[l];
    char buf[96];
    snprintf(buf, sizeof(buf), "scratch.refinenet%d.out_conv.", r);
    const std::string oc = buf;
    const uint16_t* dprow;
    if (l == 0) dprow = dpath0_;
    else { TRY(launch_unpad_rows(dpath_[l], rowsA_, B, 2 * H, 2 * W, F, st)); dprow = rowsA_; }
    TRY(lin_bwd(dprow, B * 4 * H * W, F, F, up_[l], R.out_conv.wt, rowsB_, grad(oc + "weight", (size_t)F * F), grad(oc + "bias", F), acc, st));
    TRY(launch_upsample2x_nhwc_backward(rowsB_, dmapA_[l], B, H, W, F, img_dt_, st));
    const uint16_t* in2 = R.has_u1 ? sum_[l] : rn_[l];
    TRY(rcu_backward(dmapA_[l], in2, R.u2, dmapB_[l], l, B, H, W, acc, st));
    const uint16_t* d_rn = dmapB_[l];
    if (R.has_u1) {
        dpath_[l + 1] = dmapB_[l];                 // output = path_{r+1} + RCU1(layer_rn): d path_{r+1} = d sum
        TRY(rcu_backward(dmapB_[l], rn_[l], R.u1, drn_[l], l, B, H, W, acc, st));
        d_rn = drn_[l];
    }
    snprintf(buf, sizeof(buf), "scratch.layer%d_rn.weight", r);
    TRY(conv_bwd(d_rn, L_[l], 0, layer_rn_[l], dL_[l], grad(buf, (size_t)F * cfg.reassemble_ch[l] * 9), B, H, W, cp_[l], F,
                 cfg.reassemble_ch[l], F, acc, st));
    return 0;
}

// act_postprocess[3:] backward (lseg_vit.py:446-523): d L_l -> d (ProjectReadout output) in lv_[l].dro
int Engine::reassemble_backward(int l, int B, int acc, hipStream_t st) {
    const lseg_config& c = cfg;
    const int D = c.dim, C = c.reassemble_ch[l], Cp = cp_[l], Mr = B * np_;
    LevelSave& v = lv_[l];
    char buf[96];
    snprintf(buf, sizeof(buf), "pretrained.act_postprocess%d.", l + 1);
    const std::string a = buf;
    const uint16_t* d_r1;
    if (c.resample_kind[l] == LSEG_RS_CONVT) {
        const int s = c.resample_k[l], N = s * s * Cp;
        TRY(launch_unpixshuf(dL_[l], rowsA_, B, gh_, gw_, s, Cp, st));                     // dG [Mr, s*s*Cp]
        TRY(lin_bwd(rowsA_, Mr, N, Cp, v.r1, rsmp_[l].wt, rowsB_, ws_dw_, nullptr, 0, st));
        TRY(launch_convT_wgrad_unpack(ws_dw_, grad(a + "4.weight", (size_t)C * C * s * s), C, Cp, s, acc, st));
        TRY(launch_colsum16(rowsA_, img_dt_, ws_stats_, Mr, N, N, st, 0, ws_det_, ws_det_n_));
        TRY(launch_fold_rows(ws_stats_, grad(a + "4.bias", C), s * s, C, Cp, acc, st));
        d_r1 = rowsB_;
    } else if (c.resample_kind[l] == LSEG_RS_CONV_S2) {
        const int Ho = lh_[l], Wo = lw_[l];
        TRY(launch_dilate2(dL_[l], ddil_, B, Ho, Wo, gh_, gw_, Cp, st));
        TRY(conv_bwd(ddil_, v.tmp, 0, rsmp_[l], dtmp_, grad(a + "4.weight", (size_t)C * C * 9), B, gh_, gw_, Cp, Cp, C, C, acc, st));
        TRY(bias_sum(dL_[l], grad(a + "4.bias", C), B * (Ho + 2) * (Wo + 2), C, Cp, acc, st));
        TRY(launch_unpad_rows(dtmp_, rowsB_, B, gh_, gw_, Cp, st));
        d_r1 = rowsB_;
    } else {
        TRY(launch_unpad_rows(dL_[l], rowsB_, B, gh_, gw_, Cp, st));
        d_r1 = rowsB_;
    }
    // 1x1 conv: the weight gradient is written straight into [C, D] (the padded rows >= C are not computed)
    TRY(lin_bwd(d_r1, Mr, Cp, D, v.ro, r1x1_[l].wt, v.dro, grad(a + "3.weight", (size_t)C * D), grad(a + "3.bias", C), acc, st, C, v.ropre));
    return 0;
}

// ProjectReadout backward at its hook: GELU', Linear, cat/cls scatter-add into the gradient of the hooked block output
int Engine::readout_backward(int l, int B, int acc, hipStream_t st) {
    const int D = cfg.dim, Mr = B * np_;
    LevelSave& v = lv_[l];
    char buf[96];
    snprintf(buf, sizeof(buf), "pretrained.act_postprocess%d.0.project.0.", l + 1);
    const std::string a = buf;
    TRY(lin_bwd(v.dro, Mr, D, 2 * D, v.cat, readout_[l].wt, rowsA_, grad(a + "weight", (size_t)D * 2 * D), grad(a + "bias", D), acc, st));
    TRY(launch_readout_cat_bwd(rowsA_, gx_, B, ntok_, D, img_dt_, st));
    g16_valid_ = false;
    return 0;
}

// one timm Block backward; gx_ (fp32) holds d(block output) on entry and d(block input) on exit
int Engine::block_backward(int i, int B, int acc, hipStream_t st) {
    const int D = cfg.dim, H = cfg.heads, M = B * ntok_;
    BlockSave& s = sv_[i];
    VitBlock& b = blocks_[i];
    char buf[96];
    snprintf(buf, sizeof(buf), "pretrained.model.blocks.%d.", i);
    const std::string p = buf;
    auto G = [&](const char* k, size_t n) { return grad(p + k, n); };
    float *dg1 = G("norm1.weight", D), *db1 = G("norm1.bias", D), *dg2 = G("norm2.weight", D), *db2 = G("norm2.bias", D);
    if (!dg1 || !db1 || !dg2 || !db2) return LSEG_ERR_INVALID;
    // x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
    if (!g16_valid_) TRY(launch_convert(gx_, DT_F32, g16_, img_dt_, (size_t)M * D, st));      // else: written by the previous LayerNorm backward
    TRY(lin_bwd(g16_, M, D, 4 * D, s.mlp, b.fc2.wt, dmlp_, G("mlp.fc2.weight", (size_t)D * 4 * D), G("mlp.fc2.bias", D), acc, st, -1, s.pre));
    TRY(lin_bwd(dmlp_, M, 4 * D, D, s.ln2, b.fc1.wt, dln_, G("mlp.fc1.weight", (size_t)4 * D * D), G("mlp.fc1.bias", 4 * D), acc, st));
    TRY(launch_layernorm_backward(dln_, img_dt_, s.xmid, b.g2, gx_, dg2, db2, M, D, 1e-6f, 1, st, acc, ws_ln_, g16_));
    // x_mid = x_in + proj(attention(qkv(LN1(x_in))))
    TRY(lin_bwd(g16_, M, D, D, s.att, b.proj.wt, datt_, G("attn.proj.weight", (size_t)D * D), G("attn.proj.bias", D), acc, st));
    TRY(launch_attention_backward_qkv(s.q, s.k, s.vt, s.att, datt_, s.lse, dqkv_, attn_ws_, B, H, ntok_, npad_, img_dt_, 0.125f, st));
    TRY(lin_bwd(dqkv_, M, 3 * D, D, s.ln1, b.qkv.wt, dln_, G("attn.qkv.weight", (size_t)3 * D * D), G("attn.qkv.bias", 3 * D), acc, st));
    TRY(launch_layernorm_backward(dln_, img_dt_, s.xin, b.g1, gx_, dg1, db1, M, D, 1e-6f, 1, st, acc, ws_ln_, g16_));
    g16_valid_ = true;          // g16_ = bf16(gx_) until something else adds into gx_ (a readout hook)
    return 0;
}

// lseg_backward: gradients of mean CE(ignore_index) (target given) or of <dlogits, logits> (dlogits given) w.r.t. every
// pretrained.* / scratch.* parameter the forward touched
// The value of the criterion on the last train-mode forward (lsegmentation_module.py:72), without the backward: mean CE pieces
// {sum of -log p[target], valid pixels} and, optionally, the pixel-accuracy counts {correct, labeled} (train_accuracy, :77-79).
// The per-pixel log-sum-exp it leaves behind is reused by a following lseg_backward on the same target.
int Engine::train_loss(const int64_t* target, int ignore_index, double* dev_loss2, int64_t* dev_counts2, hipStream_t st) {
    if (!train_mode || !train_fwd_valid_) return set_error(LSEG_ERR_STATE, "lseg_train_loss needs a train-mode lseg_forward first");
    if (!target) return set_error(LSEG_ERR_INVALID, "lseg_train_loss: target is NULL");
    LSEG_HIP_TRY(hipSetDevice(device));
    const int h1 = 2 * lh_[0], w1 = 2 * lw_[0], hw1 = h1 * w1;
    TRY(launch_seg_stats_ex(low_, target, train_B_, K_, 4 * hw1, ignore_index, counts_, nll_, nullptr, 1, h1, w1, st, lse_px_));
    if (dev_loss2) LSEG_HIP_TRY(hipMemcpyAsync(dev_loss2, nll_, 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (dev_counts2) LSEG_HIP_TRY(hipMemcpyAsync(dev_counts2, counts_, 2 * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    loss_target_ = target; loss_ignore_ = ignore_index;
    return 0;
}

int Engine::backward(const float* dlogits, const int64_t* target, int ignore_index, int acc, double* dev_loss2, hipStream_t st,
                     const float* dev_grad_scale) {
    if (!train_mode || !train_fwd_valid_) return set_error(LSEG_ERR_STATE, "lseg_backward needs a train-mode lseg_forward first");
    if (dlogits && dev_grad_scale) return set_error(LSEG_ERR_INVALID, "lseg_backward: a gradient scale only applies to the fused loss (target given)");
    if (!dlogits && !target) return set_error(LSEG_ERR_INVALID, "lseg_backward: give d(logits) or a target mask");
    LSEG_HIP_TRY(hipSetDevice(device));
    const lseg_config& c = cfg;
    const int B = train_B_, D = c.dim, F = c.features, M = B * ntok_, Mr = B * np_;
    const int h1 = 2 * lh_[0], w1 = 2 * lw_[0], hw1 = h1 * w1, Mp1 = B * hw1, Kp = (int)up64(K_);
    const float logit_scale = expf(logf(1.0f / 0.07f));
    if (!acc) TRY(zero_begin(zero_bwd_, st));
    // The reference back-propagates through `logit_scale * image_features.half() @ text_features.t()` (lseg_net.py:194) in HALF precision:
    // d(logits) is cast to fp16 (the `.float()` of :196 under autograd), dA = d(logits) @ text_features is an fp16 tensor (fp32
    // accumulation), and so is logit_scale * dA.  With d(logits) ~ 1 / (valid pixels) these values live in fp16's SUBNORMAL range (step
    // 2^-24 = 6e-8): a softmax probability below ~3e-8 * N contributes nothing.  The rows, the GEMM and the scale below reproduce exactly
    // that (fp16 rows, fp16 x fp16 MFMA with fp32 accumulation, fp16 output, fp16 rounding of the scaled value) unless flags bit 1 asks
    // for the un-rounded gradient.
    const int hdt = (cfg.flags & 2) ? img_dt_ : DT_F16;
    // ---- loss + x2 upsample^T: the correlation's dY rows ---------------------------------------------------------------------
    if (!dlogits) {      // CrossEntropyLoss(ignore_index) on output_conv(low): one pass for the loss and the per-pixel log-sum-exp, one for the rows
        if (!(loss_target_ == target && loss_ignore_ == ignore_index))      // else: lseg_train_loss left nll_ / lse_px_ of this forward and target
            TRY(launch_seg_stats_ex(low_, target, B, K_, 4 * hw1, ignore_index, counts_, nll_, nullptr, 1, h1, w1, st, lse_px_));
        TRY(launch_upsample_ce_backward_rows(low_, target, lse_px_, nll_, drows_, B, K_, h1, w1, Kp, ignore_index, hdt, st, dev_grad_scale));
        if (dev_loss2) LSEG_HIP_TRY(hipMemcpyAsync(dev_loss2, nll_, 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {             // autograd hand-over: d(logits) [B,K,2h,2w] given
        LSEG_HIP_TRY(hipMemsetAsync(drows_, 0, (size_t)Mp1 * Kp * 2, st));
        TRY(launch_upsample2x_planes_backward_rows(dlogits, drows_, B, K_, h1, w1, Kp, hdt, st));
    }
    // ---- head: correlation, L2-norm, head1 -------------------------------------------------------------------------------------
    GemmArgs g;
    gemm_args_init(g);
    g.A = drows_; g.W = tnT_; g.M = Mp1; g.N = c.out_c; g.K = Kp; g.lda = Kp; g.ldw = Kp;
    g.bias = zeros_; g.C = da_; g.out_dtype = hdt; g.ldc = c.out_c; g.map_mode = MAP_LINEAR;
    TRY(launch_gemm(g, hdt, st));
    TRY(launch_l2norm_scale_backward(da_, hdt, feat_, df_, img_dt_, Mp1, c.out_c, logit_scale, st));
    TRY(lin_bwd(df_, Mp1, c.out_c, F, path_[0], head1_.wt, dpath0_, grad("scratch.head1.weight", (size_t)c.out_c * F),
                grad("scratch.head1.bias", c.out_c), acc, st));
    // ---- refinenet1..4, reassemble -------------------------------------------------------------------------------------------
    for (int r = 1; r <= 4; ++r) TRY(refine_backward(r, B, acc, st));
    for (int l = 0; l < 4; ++l) TRY(reassemble_backward(l, B, acc, st));
    bucket_done(0, st);
    // ---- ViT blocks, readouts joining at their hooks --------------------------------------------------------------------------
    LSEG_HIP_TRY(hipMemsetAsync(gx_, 0, (size_t)M * D * sizeof(float), st));
    g16_valid_ = false;
    for (int i = c.depth - 1; i >= 0; --i) {
        for (int l = 0; l < 4; ++l)
            if (c.hooks[l] == i) TRY(readout_backward(l, B, acc, st));
        TRY(block_backward(i, B, acc, st));
        if (i > 0) bucket_done(1 + (c.depth - 1 - i), st);
    }
    // ---- embedding: patch_embed wgrad, cls_token, pos_embed (through the bilinear resize) ---------------------------------------
    const std::string vm = "pretrained.model.";
    const int PK = 3 * c.patch * c.patch;
    TRY(launch_embed_bwd(gx_, dtok_, dpos_, B, ntok_, D, img_dt_, st));
    TRY(lin_bwd(dtok_, Mr, D, PK, patchA_, nullptr, nullptr, grad(vm + "patch_embed.proj.weight", (size_t)D * PK),
                grad(vm + "patch_embed.proj.bias", D), acc, st));
    const size_t npos = (size_t)(1 + c.pos_grid * c.pos_grid) * D;
    float* gpos = grad(vm + "pos_embed", npos);
    float* gcls = grad(vm + "cls_token", D);
    if (!gpos || !gcls) return LSEG_ERR_INVALID;
    if (!acc) {
        LSEG_HIP_TRY(hipMemsetAsync(gpos, 0, npos * sizeof(float), st));
        LSEG_HIP_TRY(hipMemsetAsync(gcls, 0, D * sizeof(float), st));
    }
    TRY(launch_pos_resize_bwd(dpos_, gpos, gcls, c.pos_grid, gh_, gw_, D, st));
    bucket_done(c.depth, st);
    if (!acc) TRY(zero_end(zero_bwd_));
    return 0;
}

// ---- fused SGD (torch.optim.SGD semantics; two learning-rate groups, lsegmentation_module.py:119-127,165-171) ------------------
// One table entry per trainable parameter: fp32 master (the caller's tensor), gradient, momentum (one flat allocation) and the
// engine's same-layout copies of it (`direct_`, noted by finalize()).
int Engine::build_sgd_table() {
    std::vector<SgdSeg> segs;
    size_t mom = 0;
    for (auto& kv : grads_) {
        const std::string& key = kv.first;
        auto it = bound_.find(key);
        if (it == bound_.end() || it->second.dtype != LSEG_F32 || !kv.second.ptr) continue;
        int scr;
        if (key.compare(0, 11, "pretrained.") == 0) scr = 0;
        else if (key.compare(0, 8, "scratch.") == 0) scr = 1;
        else continue;
        if (kv.second.n != it->second.numel()) return set_error(LSEG_ERR_INVALID, "gradient of '%s' does not match the parameter", key.c_str());
        SgdSeg s;
        s.w = (float*)it->second.ptr; s.g = kv.second.ptr; s.m = nullptr; s.w16 = nullptr; s.w32 = nullptr;
        s.n = kv.second.n; s.blk0 = 0; s.scratch = scr; s.vec = 0;
        auto d = direct_.find(key);
        if (d != direct_.end() && !d->second.conflict) { s.w16 = d->second.w16; s.w32 = d->second.w32; }
        mom += (s.n + 3) / 4 * 4;
        segs.push_back(s);
        sgd_keys_[key] = 1;
    }
    if (mom > mom_flat_n_) {
        if (!sgd_first_) return set_error(LSEG_ERR_STATE, "the set of trainable parameters grew after the first optimizer step");
        mom_flat_ = (float*)dalloc(mom * sizeof(float));
        if (!mom_flat_) return set_error(LSEG_ERR_HIP, "out of device memory (momentum)");
        mom_flat_n_ = mom;
    }
    size_t off = 0;
    unsigned blk = 0;
    for (auto& s : segs) {
        s.m = mom_flat_ + off;
        off += (s.n + 3) / 4 * 4;
        s.blk0 = blk;
        blk += (unsigned)((s.n + 4095) / 4096);
        const uintptr_t a = (uintptr_t)s.w | (uintptr_t)s.g | (uintptr_t)s.m | (uintptr_t)s.w32;
        s.vec = !(a & 15) && !((uintptr_t)s.w16 & 7) && !(s.n & 3);
    }
    if (sgd_table_ && (int)segs.size() > sgd_nseg_) sgd_table_ = nullptr;      // (rebuilt after a re-bind; the old table stays in the arena)
    if (!sgd_table_ && !segs.empty()) {
        sgd_table_ = (SgdSeg*)dalloc(segs.size() * sizeof(SgdSeg));
        if (!sgd_table_) return set_error(LSEG_ERR_HIP, "out of device memory (optimizer table)");
    }
    if (!segs.empty()) LSEG_HIP_TRY(hipMemcpy(sgd_table_, segs.data(), segs.size() * sizeof(SgdSeg), hipMemcpyHostToDevice));
    sgd_nseg_ = (int)segs.size(); sgd_blocks_ = blk;
    sgd_dirty_ = false;
    return 0;
}

// the momentum buffer of a trainable parameter (torch.optim.SGD's state['momentum_buffer']): checkpoint / resume, engine rebuilds
int Engine::sgd_momentum(const char* key, float** out, size_t* n) {
    if (sgd_dirty_) TRY(build_sgd_table());
    size_t off = 0;
    for (auto& kv : grads_) {          // same walk as build_sgd_table
        auto it = bound_.find(kv.first);
        if (it == bound_.end() || it->second.dtype != LSEG_F32 || !kv.second.ptr) continue;
        if (kv.first.compare(0, 11, "pretrained.") != 0 && kv.first.compare(0, 8, "scratch.") != 0) continue;
        if (kv.first == key) { if (out) *out = mom_flat_ + off; if (n) *n = kv.second.n; return 0; }
        off += (kv.second.n + 3) / 4 * 4;
    }
    return set_error(LSEG_ERR_MISSING_PARAM, "no momentum buffer for '%s'", key ? key : "");
}

int Engine::sgd_step(float lr_pre, float lr_scr, float mu, float wd, hipStream_t st) {
    LSEG_HIP_TRY(hipSetDevice(device));
    if (!finalized_) return set_error(LSEG_ERR_STATE, "parameters not finalised");
    if (sgd_dirty_) TRY(build_sgd_table());
    TRY(launch_sgd_multi(sgd_table_, sgd_nseg_, sgd_blocks_, lr_pre, lr_scr, mu, wd, sgd_first_ ? 1 : 0, img_dt_, st));
    sgd_first_ = false;
    // refresh the packs with a real re-layout (padded / tap-major / transposed / flipped copies) from the updated masters; the straight
    // copies were written by the optimizer kernel, the frozen text tower and the eval-only packs are left alone
    partial_pack_ = true;
    const int r = finalize(st);
    partial_pack_ = false;
    return r;
}

}  // namespace lseg

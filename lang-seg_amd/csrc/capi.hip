// capi.hip -- extern "C" surface of liblseg_hip.so (declared in include/lseg_hip.h).
#include <cstdlib>
#include <cstring>
#include <new>

#include "engine.h"

using namespace lseg;

struct lseg_engine { Engine* e; };

#define GUARD(h) if (!(h) || !(h)->e) return set_error(LSEG_ERR_INVALID, "NULL handle")

static int require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return set_error(LSEG_ERR_NO_DEVICE, "no HIP device visible: the LSeg engine has no CPU fallback");
    return 0;
}

extern "C" {

int lseg_abi_version(void) { return LSEG_ABI_VERSION; }

int lseg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* lseg_last_error(lseg_handle h) { (void)h; return last_error(); }

int lseg_create(const lseg_config* cfg, int device, lseg_handle* out) {
    if (!cfg || !out) return set_error(LSEG_ERR_INVALID, "lseg_create: NULL argument");
    int r = require_device();
    if (r) return r;
    int n = lseg_device_count();
    if (device < 0 || device >= n) return set_error(LSEG_ERR_NO_DEVICE, "device %d not in [0,%d)", device, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return set_error(LSEG_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(LSEG_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    Engine* e = new (std::nothrow) Engine(*cfg, device);
    if (!e) return set_error(LSEG_ERR_HIP, "out of host memory");
    r = e->init();
    if (r) { delete e; return r; }
    lseg_engine* h = new (std::nothrow) lseg_engine{e};
    if (!h) { delete e; return set_error(LSEG_ERR_HIP, "out of host memory"); }
    *out = h;
    return LSEG_OK;
}

int lseg_destroy(lseg_handle h) {
    if (!h) return LSEG_OK;
    delete h->e;
    delete h;
    return LSEG_OK;
}

int lseg_bind_param(lseg_handle h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim) {
    GUARD(h);
    return h->e->bind(key, dev_ptr, dtype, shape, ndim);
}
int lseg_finalize_params(lseg_handle h, void* stream) { GUARD(h); return h->e->finalize((hipStream_t)stream); }

int lseg_set_text_tokens(lseg_handle h, const int64_t* host_tokens, int K, int ctx) {
    GUARD(h);
    if (!host_tokens) return set_error(LSEG_ERR_INVALID, "tokens NULL");
    return h->e->set_tokens(host_tokens, K, ctx);
}
int lseg_set_text_features(lseg_handle h, const void* dev_feat_f16, int K, void* stream) {
    GUARD(h);
    return h->e->set_text_features(dev_feat_f16, K, (hipStream_t)stream);
}
int lseg_encode_text(lseg_handle h, void* stream) { GUARD(h); return h->e->encode_text((hipStream_t)stream); }
int lseg_set_text_cache(lseg_handle h, int enabled) { GUARD(h); h->e->text_cache = enabled != 0; return LSEG_OK; }
int lseg_set_text_grouping(lseg_handle h, int labels_per_image) {
    GUARD(h);
    if (labels_per_image < 0 || labels_per_image > h->e->cfg.max_labels)
        return set_error(LSEG_ERR_INVALID, "labels_per_image=%d outside [0, max_labels]", labels_per_image);
    h->e->group_k = labels_per_image;
    return LSEG_OK;
}
int lseg_get_text_features(lseg_handle h, void* dev_out, void* stream) {
    GUARD(h);
    if (!dev_out) return set_error(LSEG_ERR_INVALID, "out NULL");
    return h->e->get_text_features(dev_out, (hipStream_t)stream);
}

int lseg_forward(lseg_handle h, const float* dev_x, int B, float* dev_logits_out, uint8_t* dev_argmax_out, void* stream) {
    GUARD(h);
    return h->e->forward(dev_x, B, dev_logits_out, dev_argmax_out, (hipStream_t)stream);
}

int lseg_forward_stats(lseg_handle h, const int64_t* dev_target, int ignore_index, int64_t* dev_counts, double* dev_nll, void* stream) {
    GUARD(h);
    return h->e->forward_stats(dev_target, ignore_index, dev_counts, dev_nll, (hipStream_t)stream);
}

int lseg_set_debug(lseg_handle h, int enabled) { GUARD(h); h->e->debug = enabled != 0; return LSEG_OK; }
int lseg_get_intermediate(lseg_handle h, const char* name, float* dev_out, size_t cap, size_t* n, void* stream) {
    GUARD(h);
    if (!name || !dev_out) return set_error(LSEG_ERR_INVALID, "NULL argument");
    return h->e->get_intermediate(name, dev_out, cap, n, (hipStream_t)stream);
}

int lseg_check_range(lseg_handle h, uint64_t* out4, void* stream) {
    GUARD(h);
    if (!out4) return set_error(LSEG_ERR_INVALID, "NULL argument");
    unsigned long long tmp[4] = {0, 0, 0, 0};
    const int r = h->e->check_range(tmp, (hipStream_t)stream);
    for (int i = 0; i < 4; ++i) out4[i] = tmp[i];
    return r;
}

int lseg_overflow_seen(lseg_handle h, int reset) { GUARD(h); return h->e->overflow_seen(reset != 0); }

int lseg_set_profiling(lseg_handle h, int enabled) {
    GUARD(h);
    // 1 = the whole forward + the MLP fc1 GEMM (rounds 1-2); otherwise a mask, bit f = family f in lseg_get_profile's order
    h->e->prof_mask = enabled == 1 ? 3u : (unsigned)enabled;
    if (enabled) return h->e->reserve_events(2 * h->e->events_per_forward() * 64);     // events for 64 forwards, created outside any timed loop
    return LSEG_OK;
}
int lseg_get_profile(lseg_handle h, const char* family, double* total_ms, int64_t* launches, double* flops) {
    GUARD(h);
    if (!family) return set_error(LSEG_ERR_INVALID, "family NULL");
    return h->e->get_profile(family, total_ms, launches, flops);
}

// ---- training step -----------------------------------------------------------------------------------------
int lseg_set_train(lseg_handle h, int enabled) { GUARD(h); return h->e->set_train(enabled != 0); }
int lseg_bind_grad(lseg_handle h, const char* key, float* dev_grad) { GUARD(h); return h->e->bind_grad(key, dev_grad); }
int lseg_grad_ptr(lseg_handle h, const char* key, float** dev_out, size_t* n) { GUARD(h); return h->e->grad_ptr(key, dev_out, n); }
int lseg_grad_bucket(lseg_handle h, const char* key) { if (!h || !h->e || !key) return -1; return h->e->bucket_of(key); }
int lseg_num_grad_buckets(lseg_handle h) { if (!h || !h->e) return 0; return h->e->n_buckets(); }
int lseg_backward(lseg_handle h, const float* dev_dlogits, const int64_t* dev_target, int ignore_index, int accumulate,
                  double* dev_loss, void* stream) {
    GUARD(h);
    return h->e->backward(dev_dlogits, dev_target, ignore_index, accumulate, dev_loss, (hipStream_t)stream);
}
int lseg_backward_scaled(lseg_handle h, const int64_t* dev_target, int ignore_index, int accumulate, const float* dev_grad_scale, void* stream) {
    GUARD(h);
    return h->e->backward(nullptr, dev_target, ignore_index, accumulate, nullptr, (hipStream_t)stream, dev_grad_scale);
}
int lseg_train_loss(lseg_handle h, const int64_t* dev_target, int ignore_index, double* dev_loss, int64_t* dev_counts, void* stream) {
    GUARD(h);
    return h->e->train_loss(dev_target, ignore_index, dev_loss, dev_counts, (hipStream_t)stream);
}
int lseg_sgd_momentum(lseg_handle h, const char* key, float** dev_out, size_t* n) { GUARD(h); return h->e->sgd_momentum(key, dev_out, n); }
int lseg_sgd_mark_initialized(lseg_handle h, int initialized) { GUARD(h); return h->e->sgd_mark_initialized(initialized != 0); }
int lseg_set_bn_sync(lseg_handle h, lseg_reduce_cb fn, void* user, int world_size) {
    GUARD(h);
    if (world_size < 1) return set_error(LSEG_ERR_INVALID, "world_size %d", world_size);
    h->e->bn_sync_fn = fn; h->e->bn_sync_user = user; h->e->bn_world = fn ? world_size : 1;
    return LSEG_OK;
}
int lseg_set_bucket_callback(lseg_handle h, lseg_bucket_cb fn, void* user) {
    GUARD(h);
    h->e->bucket_fn = fn; h->e->bucket_user = user;
    return LSEG_OK;
}
int lseg_sgd_step(lseg_handle h, float lr_pretrained, float lr_scratch, float momentum, float weight_decay, void* stream) {
    GUARD(h);
    return h->e->sgd_step(lr_pretrained, lr_scratch, momentum, weight_decay, (hipStream_t)stream);
}

// ---- single operators -----------------------------------------------------------------------------------
static int op_dt(int lseg_dt, int* out) {
    if (lseg_dt == LSEG_F32) { *out = DT_F32; return 0; }
    if (lseg_dt == LSEG_F16) { *out = DT_F16; return 0; }
    if (lseg_dt == LSEG_BF16) { *out = DT_BF16; return 0; }
    return set_error(LSEG_ERR_INVALID, "dtype %d", lseg_dt);
}

int lseg_op_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K,
                 int ab_dtype, int out_dtype, int act, void* stream) {
    int r = require_device(); if (r) return r;
    int ab, od;
    if ((r = op_dt(ab_dtype, &ab)) || (r = op_dt(out_dtype, &od))) return r;
    GemmArgs g;
    gemm_args_init(g);
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K;
    g.bias = bias; g.act = act;
    if (residual) { g.res_mode = RES_DEST; g.res = residual; g.res_dtype = DT_F32; }
    if (getenv("LSEG_GEMM_DBG") && (atoi(getenv("LSEG_GEMM_DBG")) & 4)) {   // tools/gemm_phase_probe.py: timing dump
        g.res_mode = RES_NONE; g.res2 = residual;
    }
    g.C = C; g.out_dtype = od; g.ldc = N; g.map_mode = MAP_LINEAR;
    return launch_gemm(g, ab, (hipStream_t)stream);
}

int lseg_op_gemm_vit(const void* A, const void* W, const float* bias, void* Cq, void* Ck, void* Cv, int M, int N, int K, int ab_dtype, int kind,
                     int ntok, int npad, int max_grid, void* stream) {
    int r = require_device(); if (r) return r;
    int ab;
    if ((r = op_dt(ab_dtype, &ab))) return r;
    if (kind < 0 || kind > 3 || !bias) return set_error(LSEG_ERR_INVALID, "lseg_op_gemm_vit: kind 0..3 and a bias");
    GemmArgs g;
    gemm_args_init(g);
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K;
    g.bias = bias; g.C = Cq; g.ldc = N; g.out_dtype = ab; g.map_mode = MAP_LINEAR; g.max_grid = max_grid;
    if (kind == 1) { g.act = ACT_GELU; g.tag = 1; }
    if (kind == 2) { g.res_mode = RES_DEST; g.res = Cq; g.res_dtype = DT_F32; g.out_dtype = DT_F32; }
    if (kind == 3) {
        if (N % 192 || ntok < 1 || npad < ntok || M % ntok) return set_error(LSEG_ERR_INVALID, "lseg_op_gemm_vit: qkv needs N = 3 * heads * 64, M = B * ntok");
        g.map_mode = MAP_QKV; g.Ck = Ck; g.Cv = Cv; g.qkv_dim = N / 3; g.qkv_ntok = ntok; g.qkv_npad = npad; g.qkv_heads = N / 192;
    }
    return launch_gemm(g, ab, (hipStream_t)stream);
}

int lseg_op_gemm_res32(const void* A, const void* W, const float* bias, float* C, int M, int N, int K, int rows_alloc, int ab_dtype, int impl,
                       int max_grid, void* stream) {
    int r = require_device(); if (r) return r;
    int ab;
    if ((r = op_dt(ab_dtype, &ab))) return r;
    if (!bias || M < 1 || rows_alloc < M) return set_error(LSEG_ERR_INVALID, "lseg_op_gemm_res32: needs a bias and rows_alloc >= M");
    GemmArgs g;
    gemm_args_init(g);
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K;
    g.bias = bias; g.C = C; g.ldc = N; g.map_mode = MAP_LINEAR; g.max_grid = max_grid;
    g.res_mode = RES_DEST; g.res = C; g.res_dtype = DT_F32; g.out_dtype = DT_F32;
    g.rows_alloc = impl == 0 ? 0 : rows_alloc;
    if (impl == 1) {
        if (!gemm_res32_asm_eligible(g, ab)) return set_error(LSEG_ERR_UNSUPPORTED, "lseg_op_gemm_res32: shape / padding does not qualify for the hand-scheduled kernel");
        return launch_gemm_res32_asm(g, ab, (hipStream_t)stream);
    }
    return launch_gemm(g, ab, (hipStream_t)stream);
}

int lseg_op_colsum(const void* in, int dtype, float* out, int R, int C, int ld, int accumulate, float* det_ws, size_t det_cap, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    if (!in || !out || R < 1 || C < 1 || ld < C) return set_error(LSEG_ERR_INVALID, "lseg_op_colsum: R, C >= 1, ld >= C");
    return launch_colsum16(in, dt, out, R, C, ld, (hipStream_t)stream, accumulate, det_ws, det_ws ? det_cap : 0);
}

int lseg_op_bn_stats(const void* x, float* stats, int B, int H, int W, int C, int dtype, float* det_ws, size_t det_cap, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    if (!x || !stats || (C % 8)) return set_error(LSEG_ERR_INVALID, "lseg_op_bn_stats: C must be a multiple of 8");
    return launch_bn_stats(x, stats, B, H, W, C, dt, (hipStream_t)stream, 0, det_ws, det_ws ? det_cap : 0);
}

int lseg_op_bn_bwd_stats(const void* dy, const void* x, const float* stats, float* bstats, int B, int H, int W, int C, float eps, int dtype,
                         float* det_ws, size_t det_cap, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    if (!dy || !x || !stats || !bstats || (C % 8)) return set_error(LSEG_ERR_INVALID, "lseg_op_bn_bwd_stats: C must be a multiple of 8");
    return launch_bn_bwd_stats(dy, x, stats, bstats, B, H, W, C, eps, (double)B * H * W, dt, (hipStream_t)stream, det_ws, det_ws ? det_cap : 0);
}

int lseg_op_layernorm(const void* in, int in_dtype, const float* gamma, const float* beta, void* out, int out_dtype,
                      int M, int D, float eps, void* stream) {
    int r = require_device(); if (r) return r;
    int id, od;
    if ((r = op_dt(in_dtype, &id)) || (r = op_dt(out_dtype, &od))) return r;
    return launch_layernorm(in, id, gamma, beta, out, od, M, D, eps, (hipStream_t)stream);
}

int lseg_op_attention(const void* q, const void* k, const void* vt, void* out, int B, int H, int Ntok, int Npad,
                      int dtype, int causal, float scale, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    return launch_attention(q, k, vt, out, B, H, Ntok, Npad, dt, causal, scale, (hipStream_t)stream);
}

int lseg_op_attention_prescaled(const void* q, const void* k, const void* vt, void* out, float* lse2, int B, int H, int Ntok, int Npad,
                                int dtype, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    return launch_attention_ex(q, k, vt, out, lse2, B, H, Ntok, Npad, dt, 0, 1.0f, 1, (hipStream_t)stream);
}

int lseg_op_conv3x3(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B,
                    int H, int W, int Cin, int Cout, int stride, int relu_in, int relu_out, void* stream) {
    int r = require_device(); if (r) return r;
    GemmArgs g;
    gemm_args_init(g);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    g.A = (const uint16_t*)in; g.W = (const uint16_t*)w_packed; g.M = B * Ho * Wo; g.N = Cout; g.K = 9 * Cin;
    g.lda = Cin; g.ldw = 9 * Cin;
    g.conv = 1; g.cin = Cin; g.hp = H + 2; g.wp = W + 2; g.ho = Ho; g.wo = Wo; g.stride = stride; g.relu_in = relu_in;
    g.bias = bias; g.act = relu_out ? ACT_RELU : ACT_NONE;
    if (residual) { g.res_mode = RES_DEST; g.res = residual; g.res_dtype = DT_BF16; }
    g.C = out; g.out_dtype = DT_BF16; g.ldc = Cout; g.map_mode = MAP_PADDED;
    return launch_gemm(g, DT_BF16, (hipStream_t)stream);
}

int lseg_op_upsample2x_nhwc(const void* in, void* out, int B, int H, int W, int C, void* stream) {
    int r = require_device(); if (r) return r;
    return launch_upsample2x_nhwc(in, out, B, H, W, C, DT_BF16, (hipStream_t)stream);
}

int lseg_op_upsample2x_planes(const float* in, float* out, int P, int H, int W, void* stream) {
    int r = require_device(); if (r) return r;
    return launch_upsample2x_planes(in, out, P, H, W, (hipStream_t)stream);
}

int lseg_op_upsample4x_planes_scaled(const float* in_padded, const float* scale, float* out, int B, int K, int H, int W, int two_stage,
                                     float* low_scratch, void* stream) {
    int r = require_device(); if (r) return r;
    if (!in_padded || !scale || !out || B < 1 || K < 1 || H < 2 || W < 2) return set_error(LSEG_ERR_INVALID, "upsample4x_planes_scaled: bad arguments");
    if (!two_stage) return launch_upsample4x_planes_scaled(in_padded, scale, out, B * K, K, H, W, (hipStream_t)stream);
    if (!low_scratch) return set_error(LSEG_ERR_INVALID, "upsample4x_planes_scaled: the two-stage form needs the [B,K,2H,2W] scratch");
    r = launch_upsample2x_planes_scaled(in_padded, scale, low_scratch, B * K, K, H, W, (hipStream_t)stream);
    if (r) return r;
    return launch_upsample2x_planes(low_scratch, out, B * K, 2 * H, 2 * W, (hipStream_t)stream);
}

int lseg_op_correlation(const float* feat, const void* text_f16, float* logits, int B, int P, int C, int K,
                        float logit_scale, void* stream) {
    int r = require_device(); if (r) return r;
    if (C % 64) return set_error(LSEG_ERR_UNSUPPORTED, "correlation: C=%d must be a multiple of 64", C);
    const size_t M = (size_t)B * P;
    uint16_t* a = nullptr;
    LSEG_HIP_TRY(hipMallocAsync((void**)&a, M * C * sizeof(uint16_t), (hipStream_t)stream));
    r = launch_l2norm_scale_f16(feat, a, (int)M, C, logit_scale, (hipStream_t)stream);
    if (!r) {
        GemmArgs g;
        gemm_args_init(g);
        if (P % 4 == 0) {      // the engine's orientation: labels = rows, pixels = columns (16-byte stores along the label planes)
            g.A = (const uint16_t*)text_f16; g.W = a; g.M = K; g.N = (int)M; g.K = C; g.lda = C; g.ldw = C;
            g.round_mid = 1; g.C = logits; g.out_dtype = DT_F32; g.map_mode = MAP_LABELPLANES; g.p_div = P;
        } else {
            g.A = a; g.W = (const uint16_t*)text_f16; g.M = (int)M; g.N = K; g.K = C; g.lda = C; g.ldw = C;
            g.round_mid = 1; g.C = logits; g.out_dtype = DT_F32; g.map_mode = MAP_NCHW; g.p_div = P;
        }
        r = launch_gemm(g, DT_F16, (hipStream_t)stream);
    }
    (void)hipFreeAsync(a, (hipStream_t)stream);
    return r;
}

int lseg_op_head_features(const void* x_bf16, const void* w_bf16, const float* bias, void* a_f16, int M, int F,
                          float logit_scale, void* stream) {
    int r = require_device(); if (r) return r;
    GemmArgs g;
    gemm_args_init(g);
    g.A = (const uint16_t*)x_bf16; g.W = (const uint16_t*)w_bf16; g.M = M; g.N = 512; g.K = F; g.lda = F; g.ldw = F;
    g.bias = bias; g.C = a_f16; g.out_dtype = DT_F16; g.ldc = 512; g.map_mode = MAP_ROWNORM; g.rn_scale = logit_scale;
    return launch_gemm(g, DT_BF16, (hipStream_t)stream);
}

int lseg_op_seg_stats(const float* d_scores, const int64_t* d_target, int B, int K, int H, int W, int ignore_index,
                      int64_t* d_counts, double* d_nll, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_scores || !d_target || !d_counts || !d_nll) return set_error(LSEG_ERR_INVALID, "seg_stats: NULL pointer");
    if (B < 1 || K < 1 || K > 4096 || H < 1 || W < 1) return set_error(LSEG_ERR_INVALID, "seg_stats: bad shape B=%d K=%d %dx%d", B, K, H, W);
    return launch_seg_stats(d_scores, d_target, B, K, H * W, ignore_index, reinterpret_cast<unsigned long long*>(d_counts),
                            d_nll, (hipStream_t)stream);
}

int lseg_op_seg_stats_lowres(const float* d_low, const int64_t* d_target, int B, int K, int h, int w, int ignore_index,
                             int64_t* d_counts, double* d_nll, uint8_t* d_argmax, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_low || (!d_target && !d_argmax) || (d_target && (!d_counts || !d_nll))) return set_error(LSEG_ERR_INVALID, "seg_stats_lowres: NULL pointer");
    if (B < 1 || K < 1 || h < 2 || w < 2) return set_error(LSEG_ERR_INVALID, "seg_stats_lowres: bad shape");
    return launch_seg_stats_ex(d_low, d_target, B, K, 4 * h * w, ignore_index, reinterpret_cast<unsigned long long*>(d_counts), d_nll,
                               d_argmax, 1, h, w, (hipStream_t)stream);
}

int lseg_op_corr_planes(const void* d_g16pad, const void* d_text16, float* d_planes, float* d_gram, int B, int K, int H, int W, int C,
                        void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_g16pad || !d_text16 || !d_planes) return set_error(LSEG_ERR_INVALID, "corr_planes: NULL pointer");
    return launch_corr_planes(d_g16pad, d_text16, d_planes, d_gram, B, K, H, W, C, (hipStream_t)stream);
}

int lseg_op_upsample_ce_backward_rows(const float* d_low, const int64_t* d_target, int B, int K, int h, int w, int ignore_index,
                                      double* d_nll, float* d_lse_ws, void* d_rows, int ldk, int out_dtype, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_low || !d_target || !d_nll || !d_lse_ws || !d_rows) return set_error(LSEG_ERR_INVALID, "upsample_ce_backward_rows: NULL pointer");
    if (B < 1 || K < 1 || h < 2 || w < 2) return set_error(LSEG_ERR_INVALID, "upsample_ce_backward_rows: bad shape");
    int dt;
    if (out_dtype == LSEG_BF16) dt = DT_BF16; else if (out_dtype == LSEG_F16) dt = DT_F16;
    else return set_error(LSEG_ERR_INVALID, "upsample_ce_backward_rows: out_dtype %d", out_dtype);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* counts = nullptr;
    LSEG_HIP_TRY(hipMallocAsync((void**)&counts, (size_t)(2 + 3 * K) * sizeof(unsigned long long), st));
    r = launch_seg_stats_ex(d_low, d_target, B, K, 4 * h * w, ignore_index, counts, d_nll, nullptr, 1, h, w, st, d_lse_ws);
    if (!r) r = launch_upsample_ce_backward_rows(d_low, d_target, d_lse_ws, d_nll, d_rows, B, K, h, w, ldk, ignore_index, dt, st);
    (void)hipFreeAsync(counts, st);
    return r;
}

int lseg_op_eval_make_crops(const float* d_img, float* d_crops, int C, int height, int width, int crop, int stride, int h_grids, int w_grids,
                            int flip, const float* host_pad3, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_img || !d_crops || !host_pad3) return set_error(LSEG_ERR_INVALID, "eval_make_crops: NULL pointer");
    if (C < 1 || C > 3 || height < 1 || width < 1 || crop < 1 || stride < 1 || h_grids < 1 || w_grids < 1)
        return set_error(LSEG_ERR_INVALID, "eval_make_crops: bad geometry");
    return launch_eval_make_crops(d_img, d_crops, C, height, width, crop, stride, h_grids, w_grids, flip ? 1 : 0, host_pad3, (hipStream_t)stream);
}
int lseg_op_eval_accumulate(const float* d_outs, float* d_map, int K, int height, int width, int ph, int pw, int crop, int stride,
                            int h_grids, int w_grids, int flip, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_outs || !d_map) return set_error(LSEG_ERR_INVALID, "eval_accumulate: NULL pointer");
    if (K < 1 || height < 1 || width < 1 || height > ph || width > pw || crop < 1 || stride < 1 || h_grids < 1 || w_grids < 1 ||
        (h_grids - 1) * stride + crop < ph || (w_grids - 1) * stride + crop < pw)
        return set_error(LSEG_ERR_INVALID, "eval_accumulate: the boxes do not cover the %dx%d map", ph, pw);
    return launch_eval_accumulate(d_outs, d_map, K, height, width, ph, pw, crop, stride, h_grids, w_grids, flip ? 1 : 0, (hipStream_t)stream);
}
int lseg_op_eval_resize(const float* d_src, float* d_dst, int P, int Hi, int Wi, int Ho, int Wo, int accumulate, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_src || !d_dst) return set_error(LSEG_ERR_INVALID, "eval_resize: NULL pointer");
    if (P < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return set_error(LSEG_ERR_INVALID, "eval_resize: bad shape");
    return launch_eval_resize(d_src, d_dst, P, Hi, Wi, Ho, Wo, accumulate ? 1 : 0, (hipStream_t)stream);
}

int lseg_op_linear_backward(const void* d_dy, const void* d_x, const void* d_w, int ab_dtype, void* d_dx, float* d_dw,
                            float* d_db, int M, int N, int K, void* stream) {
    int r = require_device(); if (r) return r;
    int ab;
    if ((r = op_dt(ab_dtype, &ab))) return r;
    if (ab != DT_BF16 && ab != DT_F16) return set_error(LSEG_ERR_INVALID, "linear_backward: operands must be bf16 or fp16");
    if (!d_dy || !d_x || !d_w) return set_error(LSEG_ERR_INVALID, "linear_backward: NULL operand");
    if (M < 1 || (N % 64) || (K % 64) || N < 64 || K < 64)
        return set_error(LSEG_ERR_UNSUPPORTED, "linear_backward: N=%d and K=%d must be multiples of 64 (M=%d >= 1)", N, K, M);
    hipStream_t st = (hipStream_t)stream;
    const int Mp = (M + 63) / 64 * 64;                    // wgrad contracts over M: pad to the GEMM's K-step with zeros
    uint16_t *wt = nullptr, *dyt = nullptr, *xt = nullptr;
    auto cleanup = [&]() { if (wt) (void)hipFreeAsync(wt, st); if (dyt) (void)hipFreeAsync(dyt, st); if (xt) (void)hipFreeAsync(xt, st); };
    GemmArgs g;
    if (d_dx) {     // dX[M,K] = dY[M,N] . W[N,K]      (A = dY, "weights" = W^T [K,N], contraction over N)
        if (hipMallocAsync((void**)&wt, (size_t)K * N * 2, st) != hipSuccess) { cleanup(); return set_error(LSEG_ERR_HIP, "linear_backward: out of device memory"); }
        if ((r = launch_transpose16(d_w, wt, N, K, K, N, st))) { cleanup(); return r; }
        gemm_args_init(g);
        g.A = (const uint16_t*)d_dy; g.W = wt; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldw = N;
        g.C = d_dx; g.out_dtype = ab; g.ldc = K; g.map_mode = MAP_LINEAR;
        if ((r = launch_gemm(g, ab, st))) { cleanup(); return r; }
    }
    float* wsk = nullptr;
    if (d_dw && wgrad_kmajor_ok(N, K, (size_t)16 * N * K)) {     // the operands as they are (K-major path: no transposed copies)
        if (hipMallocAsync((void**)&wsk, (size_t)16 * N * K * sizeof(float), st) != hipSuccess) { cleanup(); return set_error(LSEG_ERR_HIP, "linear_backward: out of device memory"); }
        r = launch_wgrad_kmajor(d_dy, N, d_x, K, M, N, K, d_dw, 0, wsk, (size_t)16 * N * K, ab, st);
        (void)hipFreeAsync(wsk, st);
        if (r) { cleanup(); return r; }
    } else if (d_dw) {     // dW[N,K] = dY^T[N,M] . X[M,K]    (A = dY^T [N,Mp], "weights" = X^T [K,Mp], contraction over M)
        if (hipMallocAsync((void**)&dyt, (size_t)N * Mp * 2, st) != hipSuccess ||
            hipMallocAsync((void**)&xt, (size_t)K * Mp * 2, st) != hipSuccess) { cleanup(); return set_error(LSEG_ERR_HIP, "linear_backward: out of device memory"); }
        if ((r = launch_transpose16(d_dy, dyt, M, N, N, Mp, st)) || (r = launch_transpose16(d_x, xt, M, K, K, Mp, st))) { cleanup(); return r; }
        gemm_args_init(g);
        g.A = dyt; g.W = xt; g.M = N; g.N = K; g.K = Mp; g.lda = Mp; g.ldw = Mp;
        g.C = d_dw; g.out_dtype = DT_F32; g.ldc = K; g.map_mode = MAP_LINEAR;
        if ((r = launch_gemm(g, ab, st))) { cleanup(); return r; }
    }
    if (d_db && (r = launch_colsum16(d_dy, ab, d_db, M, N, N, st))) { cleanup(); return r; }
    cleanup();
    return LSEG_OK;
}

int lseg_op_layernorm_backward(const void* d_dy, int dy_dtype, const float* d_x, const float* d_gamma, float* d_dx,
                               float* d_dgamma, float* d_dbeta, int M, int D, float eps, int accumulate_dx, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dy_dtype, &dt))) return r;
    if (!d_dy || !d_x || !d_gamma || !d_dx || !d_dgamma || !d_dbeta) return set_error(LSEG_ERR_INVALID, "layernorm_backward: NULL pointer");
    if (M < 1) return set_error(LSEG_ERR_INVALID, "layernorm_backward: M=%d", M);
    return launch_layernorm_backward(d_dy, dt, d_x, d_gamma, d_dx, d_dgamma, d_dbeta, M, D, eps, accumulate_dx, (hipStream_t)stream);
}

int lseg_op_conv3x3_backward(const void* d_dy_pad, const void* d_x_pad, const void* d_w_packed, void* d_dx_pad, float* d_dw,
                             int B, int H, int W, int Cin, int Cout, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_dy_pad || !d_w_packed) return set_error(LSEG_ERR_INVALID, "conv3x3_backward: NULL operand");
    if ((Cin % 64) || (Cout % 64)) return set_error(LSEG_ERR_UNSUPPORTED, "conv3x3_backward: Cin=%d, Cout=%d must be multiples of 64", Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    uint16_t *wd = nullptr, *dyt = nullptr, *xt9 = nullptr;
    auto cleanup = [&]() { if (wd) (void)hipFreeAsync(wd, st); if (dyt) (void)hipFreeAsync(dyt, st); if (xt9) (void)hipFreeAsync(xt9, st); };
    auto fail = [&](int code) { cleanup(); return code; };
    if (d_dx_pad) {
        // dX = conv3x3(dY, W flipped and channel-swapped): the forward implicit-GEMM kernel, roles of Cin / Cout exchanged
        if (hipMallocAsync((void**)&wd, (size_t)Cin * 9 * Cout * 2, st) != hipSuccess) return fail(set_error(LSEG_ERR_HIP, "conv3x3_backward: out of device memory"));
        if ((r = launch_conv_dgrad_pack(d_w_packed, wd, Cout, Cin, st))) return fail(r);
        if ((r = lseg_op_conv3x3(d_dy_pad, wd, nullptr, nullptr, d_dx_pad, B, H, W, Cout, Cin, 1, 0, 0, stream))) return fail(r);
    }
    if (d_dw) {
        // dW[co, tap, ci] = sum over padded positions m of dY[m, co] * X[m + shift(tap), ci]; dY's zero border removes the
        // positions whose shifted partner falls outside the image.  One GEMM: dY^T [Co, Mp] x (9 shifted X^T) [9*Ci, Mp]^T.
        if (!d_x_pad) return fail(set_error(LSEG_ERR_INVALID, "conv3x3_backward: wgrad needs the forward input"));
        const int Mp = B * (H + 2) * (W + 2), Mpp = (Mp + 63) / 64 * 64;
        const size_t wsn = (size_t)16 * Cout * 9 * Cin;
        if ((Cin % 128) == 0 && wgrad_kmajor_ok(Cout, 9 * Cin, wsn)) {      // the maps as they are (K-major operands, tap shifts in the loader)
            float* wsk = nullptr;
            int ns = 1;
            if (hipMallocAsync((void**)&wsk, wsn * sizeof(float), st) != hipSuccess) return fail(set_error(LSEG_ERR_HIP, "conv3x3_backward: out of device memory"));
            r = launch_conv_wgrad_kmajor(d_dy_pad, d_x_pad, 0, B, H, W, Cin, Cout, wsk, wsn, DT_BF16, &ns, st);
            if (!r) r = launch_sum_partials(wsk, d_dw, ns, (size_t)Cout * 9 * Cin, (size_t)Cout * 9 * Cin, 0, st);
            (void)hipFreeAsync(wsk, st);
            cleanup();
            return r;
        }
        if (hipMallocAsync((void**)&dyt, (size_t)Cout * Mpp * 2, st) != hipSuccess ||
            hipMallocAsync((void**)&xt9, (size_t)9 * Cin * Mpp * 2, st) != hipSuccess) return fail(set_error(LSEG_ERR_HIP, "conv3x3_backward: out of device memory"));
        if ((r = launch_transpose16(d_dy_pad, dyt, Mp, Cout, Cout, Mpp, st))) return fail(r);
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * (W + 2) + (t % 3 - 1);
            if ((r = launch_transpose16(d_x_pad, xt9 + (size_t)t * Cin * Mpp, Mp, Cin, Cin, Mpp, st, shift))) return fail(r);
        }
        GemmArgs g;
        gemm_args_init(g);
        g.A = dyt; g.W = xt9; g.M = Cout; g.N = 9 * Cin; g.K = Mpp; g.lda = Mpp; g.ldw = Mpp;
        g.C = d_dw; g.out_dtype = DT_F32; g.ldc = 9 * Cin; g.map_mode = MAP_LINEAR;
        if ((r = launch_gemm(g, DT_BF16, st))) return fail(r);
    }
    cleanup();
    return LSEG_OK;
}

int lseg_op_quickgelu_backward(const void* d_dy, const void* d_pre, void* d_dx, int64_t n, int dtype, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    if (dt == DT_F32 || !d_dy || !d_pre || !d_dx || n < 1) return set_error(LSEG_ERR_INVALID, "quickgelu_backward: bf16/fp16 tensors, n >= 1");
    return launch_gelu_backward(d_dy, d_pre, d_dx, (size_t)n, dt, (hipStream_t)stream, 1);
}

int lseg_op_gelu_backward(const void* d_dy, const void* d_pre, void* d_dx, int64_t n, int dtype, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    if (dt == DT_F32 || !d_dy || !d_pre || !d_dx || n < 1) return set_error(LSEG_ERR_INVALID, "gelu_backward: bf16/fp16 tensors, n >= 1");
    return launch_gelu_backward(d_dy, d_pre, d_dx, (size_t)n, dt, (hipStream_t)stream);
}

int lseg_op_upsample2x_nhwc_backward(const void* d_dout, void* d_din_pad, int B, int H, int W, int C, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_dout || !d_din_pad || B < 1 || H < 2 || W < 2) return set_error(LSEG_ERR_INVALID, "upsample2x_nhwc_backward: bad arguments");
    return launch_upsample2x_nhwc_backward(d_dout, d_din_pad, B, H, W, C, DT_BF16, (hipStream_t)stream);
}

int lseg_op_softmax_ce_backward(const float* d_scores, const int64_t* d_target, float* d_dscores, int B, int K, int H, int W,
                                int ignore_index, const double* d_nll, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_scores || !d_target || !d_dscores || !d_nll || B < 1 || K < 1) return set_error(LSEG_ERR_INVALID, "softmax_ce_backward: bad arguments");
    return launch_softmax_ce_backward(d_scores, d_target, d_dscores, B, K, H * W, ignore_index, d_nll, (hipStream_t)stream);
}

size_t lseg_op_attention_backward_ws_bytes(int B, int H, int Npad) { return attention_backward_ws_bytes(B, H, Npad); }

int lseg_op_attention_backward_qkv(const void* d_q, const void* d_k, const void* d_vt, const void* d_o, const void* d_do,
                                   const float* d_lse2, void* d_dqkv, void* d_ws, int B, int H, int Ntok, int Npad, int dtype,
                                   float scale, void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(dtype, &dt))) return r;
    if (!d_q || !d_k || !d_vt || !d_o || !d_do || !d_lse2 || !d_dqkv) return set_error(LSEG_ERR_INVALID, "attention_backward_qkv: NULL pointer");
    if (B < 1 || H < 1 || Ntok < 1) return set_error(LSEG_ERR_INVALID, "attention_backward_qkv: bad shape");
    hipStream_t st = (hipStream_t)stream;
    void* ws = d_ws;
    if (!ws && hipMallocAsync(&ws, attention_backward_ws_bytes(B, H, Npad), st) != hipSuccess)
        return set_error(LSEG_ERR_HIP, "attention_backward_qkv: out of device memory");
    r = launch_attention_backward_qkv(d_q, d_k, d_vt, d_o, d_do, d_lse2, d_dqkv, ws, B, H, Ntok, Npad, dt, scale, st);
    if (!d_ws) (void)hipFreeAsync(ws, st);
    return r;
}

int lseg_op_bn_train_forward(const void* d_x_pad, void* d_y_pad, float* d_stats, const float* d_gamma, const float* d_beta,
                             int B, int H, int W, int C, float eps, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_x_pad || !d_stats || (d_y_pad && (!d_gamma || !d_beta))) return set_error(LSEG_ERR_INVALID, "bn_train_forward: NULL pointer");
    return launch_bn_train_forward(d_x_pad, d_y_pad, d_stats, d_gamma, d_beta, B, H, W, C, eps, DT_BF16, (hipStream_t)stream);
}

int lseg_op_bn_train_backward(const void* d_dy_pad, const void* d_x_pad, const float* d_stats, const float* d_gamma, void* d_dx_pad,
                              float* d_dgamma_dbeta, int B, int H, int W, int C, float eps, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_dy_pad || !d_x_pad || !d_stats || !d_gamma || !d_dx_pad || !d_dgamma_dbeta) return set_error(LSEG_ERR_INVALID, "bn_train_backward: NULL pointer");
    return launch_bn_train_backward(d_dy_pad, d_x_pad, d_stats, d_gamma, d_dx_pad, d_dgamma_dbeta, B, H, W, C, eps, DT_BF16, (hipStream_t)stream);
}

int lseg_op_relu_backward(const void* d_dy, const void* d_x, void* d_dx, int64_t n, void* stream) {
    int r = require_device(); if (r) return r;
    if (!d_dy || !d_x || !d_dx || n < 1) return set_error(LSEG_ERR_INVALID, "relu_backward: bad arguments");
    return launch_relu_backward(d_dy, d_x, d_dx, (size_t)n, (hipStream_t)stream);
}

int lseg_op_upsample2x_planes_backward_rows(const float* d_dout, void* d_rows, int B, int K, int H, int W, int ldk, int out_dtype,
                                            void* stream) {
    int r = require_device(); if (r) return r;
    int dt;
    if ((r = op_dt(out_dtype, &dt))) return r;
    if (dt == DT_F32 || !d_dout || !d_rows || B < 1 || K < 1 || H < 2 || W < 2) return set_error(LSEG_ERR_INVALID, "upsample2x_planes_backward_rows: bad arguments");
    return launch_upsample2x_planes_backward_rows(d_dout, d_rows, B, K, H, W, ldk, dt, (hipStream_t)stream);
}

int lseg_op_l2norm_scale_backward(const void* d_da, int da_dtype, const float* d_x, void* d_dx, int dx_dtype, int M, int C, float scale,
                                  void* stream) {
    int r = require_device(); if (r) return r;
    int a, b;
    if ((r = op_dt(da_dtype, &a)) || (r = op_dt(dx_dtype, &b))) return r;
    if (a == DT_F32 || b == DT_F32 || !d_da || !d_x || !d_dx || M < 1) return set_error(LSEG_ERR_INVALID, "l2norm_scale_backward: 16-bit da/dx, fp32 x");
    return launch_l2norm_scale_backward(d_da, a, d_x, d_dx, b, M, C, scale, (hipStream_t)stream);
}

}  // extern "C"

// engine.hip -- whole-forward plan: LSeg.forward (modules/models/lseg_net.py:160-205) as a fixed
// sequence of gfx950 kernels over engine-owned HBM buffers.
//
// Data layout in HBM:
//   residual stream x      fp32 [B*ntok, D]                    (timm blocks keep fp32 semantics)
//   MFMA operands          bf16 (image tower) / fp16 (CLIP tower, as the reference runs it)
//   q,k                    [B*H, Npad, 64] ; v transposed [B*H, 64, Npad]  (written by the QKV GEMM)
//   DPT feature maps       NHWC 16-bit with a 1-pixel ZERO border ("padded NHWC"): the implicit-
//                          GEMM 3x3 conv reads its 9 taps with pure address arithmetic, no bounds
//                          checks; borders are zeroed once at allocation and never written.
//   pixel features         fp32 [B*h*w, out_c] -> fp16 operand a = fp16(s*fp16(f/|f|)) -> logits
//                          fp32 [B, K, h*w] (label-major planes) -> x2 bilinear -> [B,K,H,W]
#include "engine.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace lseg {

// ---- thread-local error string ------------------------------------------------------------------
static thread_local char g_err[1024] = "";
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int set_error_hip(hipError_t e, const char* what, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return LSEG_ERR_HIP;
}
const char* last_error() { return g_err; }

#define TRY(expr) do { int _r = (expr); if (_r != 0) return _r; } while (0)

static inline int T_code(int lseg_dt) { return lseg_dt == LSEG_F16 ? DT_F16 : DT_BF16; }

Engine::Engine(const lseg_config& c, int dev) : cfg(c), device(dev) {}

Engine::~Engine() {
    (void)hipSetDevice(device);
    if (text_stream_) (void)hipStreamDestroy(text_stream_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (ev_text_done_) (void)hipEventDestroy(ev_text_done_);
    if (ev_ovf_) (void)hipEventDestroy(ev_ovf_);
    if (ovf_host_) (void)hipHostFree(ovf_host_);
    for (auto e : ev_free_) (void)hipEventDestroy(e);
    for (void* p : allocs_) (void)hipFree(p);
    for (auto e : ev_pool_) (void)hipEventDestroy(e);
}

void* Engine::dalloc(size_t bytes, bool zero) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
    allocs_.push_back(p);
    return p;
}

#define ALLOC(ptr, type, count)                                                                  \
    do {                                                                                         \
        ptr = (type*)dalloc((size_t)(count) * sizeof(type));                                      \
        if (!ptr) return set_error(LSEG_ERR_HIP, "hipMalloc of %zu bytes failed (" #ptr ")",     \
                                   (size_t)(count) * sizeof(type));                              \
    } while (0)
// 16-bit image-tower tensor: in split-precision mode a (hi, lo) pair of planes, `count` elements apart (recorded in plane_)
#define ALLOC16(ptr, count)                                                                      \
    do {                                                                                         \
        ALLOC(ptr, uint16_t, (size_t)(count) * (strict_ ? 2 : 1));                                \
        if (strict_) plane_[ptr] = (size_t)(count);                                              \
        act16_.emplace_back(ptr, (size_t)(count) * (strict_ ? 2 : 1));                            \
    } while (0)

int Engine::init() {
    const lseg_config& c = cfg;
    if (c.abi_version != LSEG_ABI_VERSION) return set_error(LSEG_ERR_INVALID, "config abi_version %d != %d", c.abi_version, LSEG_ABI_VERSION);
    if (c.dim % c.heads || c.dim / c.heads != 64) return set_error(LSEG_ERR_UNSUPPORTED, "image head_dim must be 64 (dim %d / heads %d)", c.dim, c.heads);
    if (c.text_width % c.text_heads || c.text_width / c.text_heads != 64) return set_error(LSEG_ERR_UNSUPPORTED, "text head_dim must be 64");
    if (c.img_h % c.patch || c.img_w % c.patch) return set_error(LSEG_ERR_INVALID, "image size %dx%d must be a multiple of the patch size %d", c.img_h, c.img_w, c.patch);
    if (c.dim % 64 || c.features % 64 || c.out_c % 64 || c.text_width % 64 || (3 * c.patch * c.patch) % 64)
        return set_error(LSEG_ERR_UNSUPPORTED, "dim/features/out_c/text_width must be multiples of 64");
    for (int l = 0; l < 4; ++l) {
        if (c.reassemble_ch[l] % 8) return set_error(LSEG_ERR_UNSUPPORTED, "reassemble channels %d (level %d) must be a multiple of 8", c.reassemble_ch[l], l + 1);
        cp_[l] = ((c.reassemble_ch[l] + 63) / 64) * 64;      // K of the MFMA GEMMs must be a multiple of 64
    }
    if (c.max_batch < 1 || c.max_labels < 1) return set_error(LSEG_ERR_INVALID, "max_batch/max_labels");
    LSEG_HIP_TRY(hipSetDevice(device));

    strict_ = c.image_dtype == LSEG_F16_SPLIT;
    img_dt_ = strict_ ? DT_F16 : T_code(c.image_dtype);
    gh_ = c.img_h / c.patch; gw_ = c.img_w / c.patch; np_ = gh_ * gw_; ntok_ = np_ + 1;
    npad_ = ((ntok_ + 127) / 128) * 128;
    tnpad_ = ((c.text_ctx + 127) / 128) * 128;
    for (int l = 0; l < 4; ++l) {
        if (c.resample_kind[l] == LSEG_RS_CONVT) { lh_[l] = gh_ * c.resample_k[l]; lw_[l] = gw_ * c.resample_k[l]; }
        else if (c.resample_kind[l] == LSEG_RS_IDENTITY) { lh_[l] = gh_; lw_[l] = gw_; }
        else { lh_[l] = (gh_ - 1) / 2 + 1; lw_[l] = (gw_ - 1) / 2 + 1; }
    }
    // the fusion pyramid needs level l to be exactly 2x level l+1 (lseg_blocks.py:345-354)
    for (int l = 0; l < 3; ++l)
        if (lh_[l] != 2 * lh_[l + 1] || lw_[l] != 2 * lw_[l + 1])
            return set_error(LSEG_ERR_UNSUPPORTED, "reassemble pyramid is not a x2 ladder at level %d (%dx%d vs %dx%d)", l + 1, lh_[l], lw_[l], lh_[l + 1], lw_[l + 1]);
    if (2 * lh_[0] * 2 != c.img_h || 2 * lw_[0] * 2 != c.img_w)
        return set_error(LSEG_ERR_UNSUPPORTED, "head resolution %dx%d is not img/2", 2 * lh_[0], 2 * lw_[0]);

    const size_t B = c.max_batch, D = c.dim, F = c.features;
    const size_t M = B * ntok_;
    // the token-major buffers of the ViT blocks reach the next multiple of 256 rows: the hand-scheduled residual GEMM (gemm_asm.hip) works on
    // whole 256-row tiles without masks -- the padding rows are zero-initialised, computed along and never read
    const size_t Mp = (M + 255) / 256 * 256;
    rows_alloc_ = (int)Mp;
    ALLOC(x_, float, Mp * D);
    // split-K partial slabs of the residual GEMMs / deep convs at small batches (forward(): "split-K"): ns K-ranges of M rows each.  16 384
    // rows (64 MB at D = 1024) hold 4 ranges at B = 4 -- what Engine::split_residual's cost model asks for there (256 x 256 tiles for
    // mlp.fc2); round 4's 8 192 rows capped B = 4 at 2 ranges.  Measured, lease F of round 5 (tools/step_probe.py, fp16, two interleaved
    // rounds): B = 4 633 -> 654 img/s, B = 6 721 -> 718, B = 2 / 8 / 12 unchanged.  LSEG_SPLIT_ROWS: A/B switch (tools)
    static const long split_rows_env = getenv("LSEG_SPLIT_ROWS") ? atol(getenv("LSEG_SPLIT_ROWS")) : 0;
    ws_split_rows_ = split_rows_env > 0 ? (size_t)split_rows_env : 16384;
    ALLOC(ws_split_, float, ws_split_rows_ * D);
    ALLOC16(ln_, Mp * D);
    ALLOC16(q_, B * c.heads * npad_ * 64);
    ALLOC16(k_, B * c.heads * npad_ * 64);
    ALLOC16(vt_, B * c.heads * 64 * npad_);
    ALLOC16(att_, Mp * D);
    ALLOC16(mlp_, Mp * 4 * D);
    ALLOC16(patchA_, B * np_ * 3 * c.patch * c.patch);
    ALLOC16(catA_, B * np_ * 2 * D);
    ALLOC16(ro_, B * np_ * D);
    size_t r1max = 0;
    for (int l = 0; l < 4; ++l) r1max = std::max(r1max, (size_t)cp_[l]);
    ALLOC16(r1_, B * np_ * r1max);
    ALLOC16(tmp_pad_, B * (gh_ + 2) * (gw_ + 2) * r1max);
    for (int l = 0; l < 4; ++l) {
        const size_t pp = B * (lh_[l] + 2) * (lw_[l] + 2);
        ALLOC16(L_[l], pp * cp_[l]);
        ALLOC16(rn_[l], pp * F);
        if (!strict_) { ALLOC(rnr_[l], uint16_t, pp * F); ALLOC(sumr_[l], uint16_t, pp * F); }      // ReLU copies (zero border like every padded map)
        ALLOC16(t1_[l], pp * F);
        ALLOC16(sum_[l], pp * F);
        ALLOC16(t2_[l], pp * F);
        ALLOC16(up_[l], B * 4 * lh_[l] * lw_[l] * F);
        if (l > 0) ALLOC16(path_[l], B * (2 * lh_[l] + 2) * (2 * lw_[l] + 2) * F);   // padded, feeds level l-1
        else ALLOC16(path_[l], B * 4 * lh_[0] * lw_[0] * F);                         // path_1: plain rows
    }
    if (strict_) ALLOC16(relu_tmp_, B * (lh_[0] + 2) * (lw_[0] + 2) * F);
    const size_t hw1 = (size_t)4 * lh_[0] * lw_[0];
    ALLOC(gpad_, float, B * (lh_[0] + 2) * (lw_[0] + 2) * c.out_c);
    // commuted correlation: g in fp16, the label planes R = T g^T at the quarter resolution, cell dot products, per-pixel scale
    ALLOC(g16pad_, uint16_t, B * (lh_[0] + 2) * (lw_[0] + 2) * c.out_c);
    act16_.emplace_back(g16pad_, B * (lh_[0] + 2) * (lw_[0] + 2) * c.out_c);       // fp16 whatever the operand type: the correlation's operand
    ALLOC(range_out_, unsigned long long, 4);
    ALLOC(ovf_dev_, unsigned, 4);
    LSEG_HIP_TRY(hipHostMalloc((void**)&ovf_host_, 16, hipHostMallocDefault));
    ovf_host_[0] = 0;
    LSEG_HIP_TRY(hipEventCreateWithFlags(&ev_ovf_, hipEventDisableTiming));
    ALLOC(rpl_, float, B * c.max_labels * (lh_[0] + 2) * (lw_[0] + 2));
    ALLOC(gram_, float, B * lh_[0] * lw_[0] * 5);
    ALLOC(nscale_, float, B * hw1);
    ALLOC(feat_, float, B * hw1 * c.out_c);
    ALLOC(a16_, uint16_t, B * hw1 * c.out_c);
    ALLOC(low_, float, B * c.max_labels * hw1);
    if (c.arch_option == 1 || c.arch_option == 2) {
        ALLOC(low2_, float, B * c.max_labels * hw1);
        ALLOC(low3_, float, B * c.max_labels * hw1);
    }
    // text
    const size_t Kl = c.max_labels, L = c.text_ctx, W = c.text_width;
    ALLOC(d_tok_, int64_t, Kl * L);
    ALLOC(d_eot_, int, Kl);
    ALLOC(tx_, uint16_t, Kl * L * W);
    ALLOC(tln_, uint16_t, Kl * L * W);
    ALLOC(tq_, uint16_t, Kl * c.text_heads * tnpad_ * 64);
    ALLOC(tk_, uint16_t, Kl * c.text_heads * tnpad_ * 64);
    ALLOC(tvt_, uint16_t, Kl * c.text_heads * 64 * tnpad_);
    ALLOC(tatt_, uint16_t, Kl * L * W);
    ALLOC(tmlp_, uint16_t, Kl * L * 4 * W);
    ALLOC(tpool_, uint16_t, Kl * W);
    ALLOC(tfeat_, uint16_t, Kl * c.out_c);
    ALLOC(tnorm_, uint16_t, Kl * c.out_c);
    {   // the text tower's ~90 latency-bound launches run beside the image tower: at the LOWEST stream priority, so that the dispatcher hands
        // free CUs to the image tower's kernels first (LSEG_TEXT_PRIO=0: default priority, tools A/B)
        static const int prio_env = getenv("LSEG_TEXT_PRIO") ? atoi(getenv("LSEG_TEXT_PRIO")) : 1;
        int least = 0, greatest = 0;
        if (prio_env && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            LSEG_HIP_TRY(hipStreamCreateWithPriority(&text_stream_, hipStreamNonBlocking, least));
        else
            LSEG_HIP_TRY(hipStreamCreateWithFlags(&text_stream_, hipStreamNonBlocking));
    }
    LSEG_HIP_TRY(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    LSEG_HIP_TRY(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    LSEG_HIP_TRY(hipEventCreateWithFlags(&ev_text_done_, hipEventDisableTiming));
    LSEG_HIP_TRY(hipDeviceSynchronize());
    inited_ = true;
    return 0;
}

int Engine::bind(const char* key, const void* p, int dtype, const int64_t* shape, int ndim) {
    if (!key || !p || ndim < 0 || ndim > 8) return set_error(LSEG_ERR_INVALID, "bind_param: bad arguments");
    if (dtype != LSEG_F32 && dtype != LSEG_F16 && dtype != LSEG_BF16 && dtype != LSEG_I64) return set_error(LSEG_ERR_INVALID, "bind_param(%s): dtype %d", key, dtype);
    BoundParam b;
    b.ptr = p; b.dtype = dtype;
    b.shape.assign(shape, shape + ndim);
    bound_[key] = b;
    finalized_ = false;
    sgd_dirty_ = true;
    return 0;
}

void Engine::note_direct(const std::string& key, uint16_t* w16, float* w32) {
    DirectDst& d = direct_[key];
    if ((w16 && d.w16 && d.w16 != w16) || (w32 && d.w32 && d.w32 != w32)) d.conflict = true;
    if (w16) d.w16 = w16;
    if (w32) d.w32 = w32;
}

int Engine::need(const std::string& key, BoundParam& out, std::initializer_list<int64_t> shape) {
    auto it = bound_.find(key);
    if (it == bound_.end()) return set_error(LSEG_ERR_MISSING_PARAM, "parameter '%s' was never bound", key.c_str());
    out = it->second;
    size_t want = 1;
    for (auto s : shape) want *= (size_t)s;
    if (out.numel() != want) return set_error(LSEG_ERR_INVALID, "parameter '%s' has %zu elements, expected %zu", key.c_str(), out.numel(), want);
    if (out.dtype == LSEG_I64) return set_error(LSEG_ERR_INVALID, "parameter '%s' is int64", key.c_str());
    return 0;
}

int Engine::pack_f32(const std::string& key, size_t n, float*& out, hipStream_t st) {
    BoundParam p;
    TRY(need(key, p, {(int64_t)n}));
    if (!out) ALLOC(out, float, n);
    if (p.dtype == LSEG_F32) note_direct(key, nullptr, out);
    if (sgd_owns(key)) return 0;               // written by the optimizer kernel
    return launch_convert(p.ptr, p.dtype, out, DT_F32, n, st);
}

int Engine::pack_tmp(size_t n, hipStream_t st) {
    if (n > pack_tmp_n_) { ALLOC(pack_tmp_, float, n); pack_tmp_n_ = n; }
    LSEG_HIP_TRY(hipMemsetAsync(pack_tmp_, 0, n * sizeof(float), st));
    return 0;
}

int Engine::igemm(GemmArgs& g, hipStream_t st) {
    if (strict_) {
        g.split = 1;
        g.a_plane = pl(g.A); g.w_plane = pl(g.W);
        if (!g.a_plane || !g.w_plane) return set_error(LSEG_ERR_STATE, "split-precision GEMM on a tensor without a lo plane");
        if (g.out_dtype != DT_F32) { g.c_plane = pl(g.C); g.ck_plane = pl(g.Ck); g.cv_plane = pl(g.Cv); }
        if (g.res && g.res_dtype != DT_F32) g.res_plane = pl(g.res);
    }
    return launch_gemm(g, img_dt_, st);
}

int Engine::pack_linear(const std::string& wkey, const std::string& bkey, int n, int k, int dt, Lin& out, hipStream_t st, bool image) {
    BoundParam w;
    TRY(need(wkey, w, {n, k}));
    const bool split = image && strict_;
    if (!out.w) { ALLOC(out.w, uint16_t, (size_t)n * k * (split ? 2 : 1)); if (split) plane_[out.w] = (size_t)n * k; }
    out.n = n; out.k = k;
    if (!split && w.dtype == LSEG_F32 && dt == img_dt_) note_direct(wkey, out.w, nullptr);
    if (split) TRY(launch_convert_split(w.ptr, w.dtype, out.w, (size_t)n * k, (size_t)n * k, st));
    else if (!sgd_owns(wkey)) TRY(launch_convert(w.ptr, w.dtype, out.w, dt, (size_t)n * k, st));
    if (!bkey.empty()) TRY(pack_f32(bkey, n, out.b, st));
    return 0;
}

int Engine::pack_conv3(const std::string& wkey, const std::string& bnp, const std::string& bias_key, int co, int ci,
                       int cop, int cip, Lin& out, hipStream_t st) {
    BoundParam w;
    TRY(need(wkey, w, {co, ci, 3, 3}));
    if (w.dtype != LSEG_F32) return set_error(LSEG_ERR_UNSUPPORTED, "'%s' must be fp32", wkey.c_str());
    const float *bw = nullptr, *bb = nullptr, *bm = nullptr, *bv = nullptr, *cb = nullptr;
    if (!bnp.empty()) {
        BoundParam a, b, m, v;
        TRY(need(bnp + ".weight", a, {co})); TRY(need(bnp + ".bias", b, {co}));
        TRY(need(bnp + ".running_mean", m, {co})); TRY(need(bnp + ".running_var", v, {co}));
        if (a.dtype != LSEG_F32 || b.dtype != LSEG_F32 || m.dtype != LSEG_F32 || v.dtype != LSEG_F32)
            return set_error(LSEG_ERR_UNSUPPORTED, "BatchNorm '%s' must be fp32", bnp.c_str());
        bw = (const float*)a.ptr; bb = (const float*)b.ptr; bm = (const float*)m.ptr; bv = (const float*)v.ptr;
    }
    if (!bias_key.empty()) {
        BoundParam b;
        TRY(need(bias_key, b, {co}));
        if (b.dtype != LSEG_F32) return set_error(LSEG_ERR_UNSUPPORTED, "'%s' must be fp32", bias_key.c_str());
        cb = (const float*)b.ptr;
    }
    const size_t nw = (size_t)cop * 9 * cip;
    if (!out.w) { ALLOC(out.w, uint16_t, nw * (strict_ ? 2 : 1)); if (strict_) plane_[out.w] = nw; }   // zero-initialised: padding stays 0
    const bool has_bias = bw || cb;
    if (!out.b) ALLOC(out.b, float, cop);          // bias-less convs (layerN_rn) get zeros: keeps them on the specialised epilogue
    out.n = cop; out.k = 9 * cip;
    if (strict_) {                                 // fold + re-layout in fp32, then split into the (hi, lo) planes
        TRY(pack_tmp(nw, st));
        TRY(launch_pack_conv3x3((const float*)w.ptr, bw, bb, bm, bv, 1e-5f, cb, pack_tmp_, has_bias ? out.b : nullptr, co, ci, cip, DT_F32, st));
        return launch_convert_split(pack_tmp_, DT_F32, out.w, nw, nw, st);
    }
    return launch_pack_conv3x3((const float*)w.ptr, bw, bb, bm, bv, 1e-5f, cb, out.w, has_bias ? out.b : nullptr, co, ci, cip, img_dt_, st);
}

int Engine::finalize(hipStream_t st) {
    if (!inited_) return set_error(LSEG_ERR_STATE, "engine not initialised");
    LSEG_HIP_TRY(hipSetDevice(device));
    const lseg_config& c = cfg;
    const int D = c.dim, F = c.features, P = c.patch;
    char buf[256];
    const std::string vm = "pretrained.model.";
    // ---- ViT --------------------------------------------------------------------------------------------
    TRY(pack_linear(vm + "patch_embed.proj.weight", vm + "patch_embed.proj.bias", D, 3 * P * P, img_dt_, patch_, st, true));
    TRY(pack_f32(vm + "cls_token", D, cls_, st));
    TRY(pack_f32(vm + "pos_embed", (size_t)(1 + c.pos_grid * c.pos_grid) * D, pos_raw_, st));
    if (!pos_) ALLOC(pos_, float, (size_t)ntok_ * D);
    TRY(launch_pos_resize(pos_raw_, pos_, c.pos_grid, gh_, gw_, D, st));     // lseg_vit.py:149-163, once
    blocks_.resize(c.depth);
    for (int i = 0; i < c.depth; ++i) {
        snprintf(buf, sizeof(buf), "%sblocks.%d.", vm.c_str(), i);
        const std::string b = buf;
        VitBlock& k = blocks_[i];
        TRY(pack_f32(b + "norm1.weight", D, k.g1, st)); TRY(pack_f32(b + "norm1.bias", D, k.b1, st));
        TRY(pack_f32(b + "norm2.weight", D, k.g2, st)); TRY(pack_f32(b + "norm2.bias", D, k.b2, st));
        TRY(pack_linear(b + "attn.qkv.weight", b + "attn.qkv.bias", 3 * D, D, img_dt_, k.qkv, st, true));
        TRY(pack_linear(b + "attn.proj.weight", b + "attn.proj.bias", D, D, img_dt_, k.proj, st, true));
        TRY(pack_linear(b + "mlp.fc1.weight", b + "mlp.fc1.bias", 4 * D, D, img_dt_, k.fc1, st, true));
        TRY(pack_linear(b + "mlp.fc2.weight", b + "mlp.fc2.bias", D, 4 * D, img_dt_, k.fc2, st, true));
    }
    // ---- readout / reassemble / layer_rn -------------------------------------------------------------
    for (int l = 0; l < 4; ++l) {
        snprintf(buf, sizeof(buf), "pretrained.act_postprocess%d.", l + 1);
        const std::string a = buf;
        const int C = c.reassemble_ch[l], Cp = cp_[l];
        TRY(pack_linear(a + "0.project.0.weight", a + "0.project.0.bias", D, 2 * D, img_dt_, readout_[l], st, true));
        {   // 1x1 conv [C, D] -> [Cp, D] (rows >= C are zero: the padded channels stay exactly 0)
            BoundParam w, b;
            TRY(need(a + "3.weight", w, {C, D}));
            TRY(need(a + "3.bias", b, {C}));
            if (!r1x1_[l].w) { ALLOC(r1x1_[l].w, uint16_t, (size_t)Cp * D * (strict_ ? 2 : 1)); if (strict_) plane_[r1x1_[l].w] = (size_t)Cp * D; }
            if (!r1x1_[l].b) ALLOC(r1x1_[l].b, float, Cp);
            r1x1_[l].n = Cp; r1x1_[l].k = D;
            if (strict_) TRY(launch_convert_split(w.ptr, w.dtype, r1x1_[l].w, (size_t)C * D, (size_t)Cp * D, st));
            else TRY(launch_convert(w.ptr, w.dtype, r1x1_[l].w, img_dt_, (size_t)C * D, st));
            TRY(launch_convert(b.ptr, b.dtype, r1x1_[l].b, DT_F32, C, st));
        }
        if (c.resample_kind[l] == LSEG_RS_CONVT) {
            const int s = c.resample_k[l];
            BoundParam w, b;
            TRY(need(a + "4.weight", w, {C, C, s, s}));
            TRY(need(a + "4.bias", b, {C}));
            if (w.dtype != LSEG_F32) return set_error(LSEG_ERR_UNSUPPORTED, "'%s4.weight' must be fp32", a.c_str());
            const size_t nwt = (size_t)s * s * Cp * Cp;
            if (!rsmp_[l].w) { ALLOC(rsmp_[l].w, uint16_t, nwt * (strict_ ? 2 : 1)); if (strict_) plane_[rsmp_[l].w] = nwt; }
            if (!rsmp_[l].b) ALLOC(rsmp_[l].b, float, Cp);
            rsmp_[l].n = s * s * Cp; rsmp_[l].k = Cp;
            if (strict_) {
                TRY(pack_tmp(nwt, st));
                TRY(launch_pack_convT((const float*)w.ptr, pack_tmp_, C, C, Cp, s, DT_F32, st));
                TRY(launch_convert_split(pack_tmp_, DT_F32, rsmp_[l].w, nwt, nwt, st));
            } else {
                TRY(launch_pack_convT((const float*)w.ptr, rsmp_[l].w, C, C, Cp, s, img_dt_, st));
            }
            TRY(launch_convert(b.ptr, b.dtype, rsmp_[l].b, DT_F32, C, st));
        } else if (c.resample_kind[l] == LSEG_RS_CONV_S2) {
            TRY(pack_conv3(a + "4.weight", "", a + "4.bias", C, C, Cp, Cp, rsmp_[l], st));
        }
        snprintf(buf, sizeof(buf), "scratch.layer%d_rn.weight", l + 1);
        TRY(pack_conv3(buf, "", "", F, C, F, Cp, layer_rn_[l], st));
    }
    // ---- refinenets + head ------------------------------------------------------------------------------
    for (int r = 1; r <= 4; ++r) {
        snprintf(buf, sizeof(buf), "scratch.refinenet%d.", r);
        const std::string p = buf;
        Refine& R = refine_[r - 1];
        TRY(pack_linear(p + "out_conv.weight", p + "out_conv.bias", F, F, img_dt_, R.out_conv, st, true));
        for (int u = 1; u <= 2; ++u) {
            if (u == 1 && r == 4) continue;      // refinenet4.resConfUnit1 never runs (lseg_net.py:176)
            if (partial_pack_) continue;         // BN-folded packs are eval-only: refreshed lazily (eval_stale_)
            const std::string q = p + "resConfUnit" + std::to_string(u) + ".";
            Rcu& U = u == 1 ? R.u1 : R.u2;
            TRY(pack_conv3(q + "conv1.weight", q + "bn1", "", F, F, F, F, U.c1, st));
            TRY(pack_conv3(q + "conv2.weight", q + "bn2", "", F, F, F, F, U.c2, st));
        }
        R.has_u1 = r != 4;
    }
    TRY(pack_linear("scratch.head1.weight", "scratch.head1.bias", c.out_c, F, img_dt_, head1_, st, true));
    if (!partial_pack_) {   // commuted head: Wc = W_head1 . W_out_conv(refinenet1), bc = W_head1 . b_out_conv + b_head1  (fp32, then packed)
        BoundParam wh, bh, wo, bo;
        TRY(need("scratch.head1.weight", wh, {c.out_c, F})); TRY(need("scratch.head1.bias", bh, {c.out_c}));
        TRY(need("scratch.refinenet1.out_conv.weight", wo, {F, F})); TRY(need("scratch.refinenet1.out_conv.bias", bo, {F}));
        if (wh.dtype == LSEG_F32 && bh.dtype == LSEG_F32 && wo.dtype == LSEG_F32 && bo.dtype == LSEG_F32) {
            if (!headc_w32_) ALLOC(headc_w32_, float, (size_t)c.out_c * F);
            if (!headc_.b) ALLOC(headc_.b, float, c.out_c);
            if (!headc_.w) ALLOC(headc_.w, uint16_t, (size_t)c.out_c * F);
            headc_.n = c.out_c; headc_.k = F;
            TRY(launch_combine_1x1((const float*)wh.ptr, (const float*)bh.ptr, (const float*)wo.ptr, (const float*)bo.ptr, headc_w32_, headc_.b,
                                   c.out_c, F, F, st));
            TRY(launch_convert(headc_w32_, DT_F32, headc_.w, img_dt_, (size_t)c.out_c * F, st));
        } else {
            headc_.n = 0;           // non-fp32 checkpoints: the two-GEMM path
        }
    }
    if (c.arch_option == 1 || c.arch_option == 2) {
        TRY(pack_f32("scratch.head_block.depthwise.depthwise.weight", 9, hb_w_, st));
        TRY(pack_f32("scratch.head_block.depthwise.depthwise.bias", 1, hb_b_, st));
    }
    if (partial_pack_) {                         // optimizer step: the text tower is frozen, nothing below changed
        eval_stale_ = true;
        return finalize_train(st);
    }
    eval_stale_ = false;
    // ---- CLIP text tower ([3P] clip/model.py) -----------------------------------------------------------
    const std::string cp = "clip_pretrained.";
    const int W = c.text_width;
    TRY(pack_f32(cp + "token_embedding.weight", (size_t)c.text_vocab * W, tok_emb_, st));
    TRY(pack_f32(cp + "positional_embedding", (size_t)c.text_ctx * W, tpos_, st));
    tblocks_.resize(c.text_layers);
    for (int i = 0; i < c.text_layers; ++i) {
        snprintf(buf, sizeof(buf), "%stransformer.resblocks.%d.", cp.c_str(), i);
        const std::string b = buf;
        TextBlock& k = tblocks_[i];
        TRY(pack_f32(b + "ln_1.weight", W, k.g1, st)); TRY(pack_f32(b + "ln_1.bias", W, k.b1, st));
        TRY(pack_f32(b + "ln_2.weight", W, k.g2, st)); TRY(pack_f32(b + "ln_2.bias", W, k.b2, st));
        TRY(pack_linear(b + "attn.in_proj_weight", b + "attn.in_proj_bias", 3 * W, W, DT_F16, k.qkv, st));
        TRY(pack_linear(b + "attn.out_proj.weight", b + "attn.out_proj.bias", W, W, DT_F16, k.out, st));
        TRY(pack_linear(b + "mlp.c_fc.weight", b + "mlp.c_fc.bias", 4 * W, W, DT_F16, k.fc, st));
        TRY(pack_linear(b + "mlp.c_proj.weight", b + "mlp.c_proj.bias", W, 4 * W, DT_F16, k.proj, st));
    }
    TRY(pack_f32(cp + "ln_final.weight", W, tlnf_g_, st));
    TRY(pack_f32(cp + "ln_final.bias", W, tlnf_b_, st));
    {
        BoundParam tp;
        TRY(need(cp + "text_projection", tp, {W, c.out_c}));
        if (!tproj_.w) ALLOC(tproj_.w, uint16_t, (size_t)W * c.out_c);
        tproj_.n = c.out_c; tproj_.k = W;
        TRY(launch_transpose_convert(tp.ptr, tp.dtype, tproj_.w, DT_F16, W, c.out_c, st));   // x @ P == x @ (P^T)^T
    }
    LSEG_HIP_TRY(hipStreamSynchronize(st));      // the caller may now free/modify its tensors
    finalized_ = true;
    text_valid = false;
    if (train_alloc_) TRY(finalize_train(st));   // transposed / flipped / un-folded packs of the training step
    return 0;
}

int Engine::set_tokens(const int64_t* tok, int K, int ctx) {
    if (!inited_) return set_error(LSEG_ERR_STATE, "engine not initialised");
    if (K < 1 || K > cfg.max_labels) return set_error(LSEG_ERR_INVALID, "K=%d outside [1, max_labels=%d]", K, cfg.max_labels);
    if (ctx != cfg.text_ctx) return set_error(LSEG_ERR_INVALID, "token context %d != %d", ctx, cfg.text_ctx);
    LSEG_HIP_TRY(hipSetDevice(device));
    std::vector<int> eot(K);
    for (int k = 0; k < K; ++k) {
        // text.argmax(dim=-1): first occurrence of the row maximum ([3P] clip/model.py encode_text)
        int best = 0;
        for (int l = 0; l < ctx; ++l) {
            const int64_t v = tok[(size_t)k * ctx + l];
            if (v < 0 || v >= cfg.text_vocab) return set_error(LSEG_ERR_INVALID, "token id %lld out of range at [%d,%d]", (long long)v, k, l);
            if (v > tok[(size_t)k * ctx + best]) best = l;
        }
        eot[k] = best;
    }
    // The text-tower kernels that read d_tok_ / d_eot_ run asynchronously (engine side stream, or the caller's stream): wait for
    // the last encode_text to finish before overwriting them, otherwise forward N could embed / pool with forward N+1's tokens.
    if (text_pending_) { LSEG_HIP_TRY(hipEventSynchronize(ev_text_done_)); text_pending_ = false; }
    LSEG_HIP_TRY(hipMemcpy(d_tok_, tok, (size_t)K * ctx * sizeof(int64_t), hipMemcpyHostToDevice));
    LSEG_HIP_TRY(hipMemcpy(d_eot_, eot.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice));
    K_ = K;
    // Exact causal truncation: with the -inf upper-triangular mask no position after a label's EOT
    // token can influence its EOT row, and every other op of the tower is row-wise, so only the
    // first max(eot)+1 positions are computed (bit-identical pooled features; flags bit 0 = the
    // reference's full 77-position schedule).
    int lmax = 0;
    for (int k = 0; k < K; ++k) lmax = std::max(lmax, eot[k] + 1);
    tl_ = (cfg.flags & 1) ? ctx : lmax;
    text_valid = false;
    text_external_ = false;
    return 0;
}

// text features computed elsewhere (`clip_pretrained.encode_text(text)` cached by the caller, a label bank of an application): fp16
// [K, out_c], not necessarily normalised; the engine applies the fp16 L2 normalisation of lseg_net.py:192 and uses them until the
// next lseg_set_text_tokens / lseg_set_text_features
int Engine::set_text_features(const void* feat, int K, hipStream_t st) {
    if (!inited_) return set_error(LSEG_ERR_STATE, "engine not initialised");
    if (!feat || K < 1 || K > cfg.max_labels) return set_error(LSEG_ERR_INVALID, "set_text_features: K=%d outside [1, max_labels=%d]", K, cfg.max_labels);
    LSEG_HIP_TRY(hipSetDevice(device));
    if (text_pending_) { LSEG_HIP_TRY(hipEventSynchronize(ev_text_done_)); text_pending_ = false; }
    LSEG_HIP_TRY(hipMemcpyAsync(tfeat_, feat, (size_t)K * cfg.out_c * 2, hipMemcpyDeviceToDevice, st));
    TRY(launch_text_l2norm(tfeat_, tnorm_, K, cfg.out_c, st));
    K_ = K;
    text_valid = true;
    text_external_ = true;
    return 0;
}

// clip_pretrained.encode_text(text) + the fp16 L2 normalisation (lseg_net.py:183,192)
int Engine::encode_text(hipStream_t st) {
    if (!finalized_) return set_error(LSEG_ERR_STATE, "parameters not finalised");
    if (K_ < 1) return set_error(LSEG_ERR_STATE, "no text tokens set");
    const lseg_config& c = cfg;
    const int L = tl_, W = c.text_width, H = c.text_heads, M = K_ * L;
    TRY(launch_text_embed(d_tok_, tok_emb_, tpos_, tx_, M, L, c.text_ctx, W, st));
    GemmArgs g;
    for (int i = 0; i < c.text_layers; ++i) {
        TextBlock& b = tblocks_[i];
        TRY(launch_layernorm(tx_, DT_F16, b.g1, b.b1, tln_, DT_F16, M, W, 1e-5f, st));
        gemm_args_init(g);
        g.A = tln_; g.W = b.qkv.w; g.M = M; g.N = 3 * W; g.K = W; g.lda = W; g.ldw = W;
        g.bias = b.qkv.b; g.out_dtype = DT_F16; g.map_mode = MAP_QKV;
        g.C = tq_; g.Ck = tk_; g.Cv = tvt_; g.qkv_dim = W; g.qkv_ntok = L; g.qkv_npad = tnpad_; g.qkv_heads = H;
        TRY(launch_gemm(g, DT_F16, st));
        TRY(launch_attention(tq_, tk_, tvt_, tatt_, K_, H, L, tnpad_, DT_F16, 1, 0.125f, st));
        gemm_args_init(g);
        g.A = tatt_; g.W = b.out.w; g.M = M; g.N = W; g.K = W; g.lda = W; g.ldw = W;
        g.bias = b.out.b; g.round_mid = 1; g.res_mode = RES_DEST; g.res = tx_; g.res_dtype = DT_F16;
        g.C = tx_; g.out_dtype = DT_F16; g.ldc = W; g.map_mode = MAP_LINEAR;
        TRY(launch_gemm(g, DT_F16, st));
        TRY(launch_layernorm(tx_, DT_F16, b.g2, b.b2, tln_, DT_F16, M, W, 1e-5f, st));
        gemm_args_init(g);
        g.A = tln_; g.W = b.fc.w; g.M = M; g.N = 4 * W; g.K = W; g.lda = W; g.ldw = W;
        g.bias = b.fc.b; g.round_mid = 1; g.act = ACT_QUICKGELU;
        g.C = tmlp_; g.out_dtype = DT_F16; g.ldc = 4 * W; g.map_mode = MAP_LINEAR;
        TRY(launch_gemm(g, DT_F16, st));
        gemm_args_init(g);
        g.A = tmlp_; g.W = b.proj.w; g.M = M; g.N = W; g.K = 4 * W; g.lda = 4 * W; g.ldw = 4 * W;
        g.bias = b.proj.b; g.round_mid = 1; g.res_mode = RES_DEST; g.res = tx_; g.res_dtype = DT_F16;
        g.C = tx_; g.out_dtype = DT_F16; g.ldc = W; g.map_mode = MAP_LINEAR;
        TRY(launch_gemm(g, DT_F16, st));
    }
    TRY(launch_layernorm(tx_, DT_F16, tlnf_g_, tlnf_b_, tln_, DT_F16, M, W, 1e-5f, st));
    TRY(launch_text_pool(tln_, d_eot_, tpool_, K_, L, W, st));
    gemm_args_init(g);
    g.A = tpool_; g.W = tproj_.w; g.M = K_; g.N = c.out_c; g.K = W; g.lda = W; g.ldw = W;
    g.C = tfeat_; g.out_dtype = DT_F16; g.ldc = c.out_c; g.map_mode = MAP_LINEAR;
    TRY(launch_gemm(g, DT_F16, st));
    TRY(launch_text_l2norm(tfeat_, tnorm_, K_, c.out_c, st));
    LSEG_HIP_TRY(hipEventRecord(ev_text_done_, st));
    text_pending_ = true;
    text_valid = true;
    return 0;
}

int Engine::conv3x3(const void* in, const Lin& w, const void* res, const void* res2, void* out, int B, int H, int W,
                    int stride, int relu_in, int relu_out, hipStream_t st, void* out_relu, bool* relu_written) {
    // in: padded NHWC [B,H+2,W+2,Cin]; out: padded NHWC [B,Ho+2,Wo+2,Cout]
    GemmArgs g;
    gemm_args_init(g);
    const int Cin = w.k / 9, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (strict_ && relu_in) {        // the ReLU-ed input is materialised: a sign test on the hi fragments alone cannot zero the lo plane
        TRY(launch_relu_split(in, pl(in), relu_tmp_, pl(relu_tmp_), (size_t)B * (H + 2) * (W + 2) * Cin, st));
        in = relu_tmp_; relu_in = 0;
    }
    g.A = (const uint16_t*)in; g.W = w.w; g.M = B * Ho * Wo; g.N = w.n; g.K = w.k; g.lda = Cin; g.ldw = w.k;
    g.conv = 1; g.cin = Cin; g.hp = H + 2; g.wp = W + 2; g.ho = Ho; g.wo = Wo; g.stride = stride; g.relu_in = relu_in;
    g.bias = w.b; g.act = relu_out ? ACT_RELU : ACT_NONE;
    if (res) { g.res_mode = RES_DEST; g.res = res; g.res_dtype = img_dt_; g.res2 = res2; }
    g.C = out; g.out_dtype = img_dt_; g.ldc = w.n; g.map_mode = MAP_PADDED;
    if (relu_written) *relu_written = false;
    // Small batches: the deep levels of the pyramid are a handful of tiles with a 36..144-step contraction (layer3_rn at B = 1: 60 tiles of
    // 64 x 64, K = 9216, 95 us on a quarter of the chip).  Split-K work items into fp32 slabs + one streaming epilogue kernel.
    static const int split_on = getenv("LSEG_SPLITK") ? atoi(getenv("LSEG_SPLITK")) : 1;
    if (split_on && !(cfg.flags & 4) && !train_mode && !strict_ && !relu_in && (w.n % 64) == 0) {
        const int nk = w.k / 64;
        const long tiles = (long)((g.M + 63) / 64) * (w.n / 64);
        long ns = 512 / tiles;
        if (ns > nk / 8) ns = nk / 8;
        if (ns > 8) ns = 8;
        while (ns > 1 && (size_t)ns * g.M * w.n > ws_split_rows_ * (size_t)cfg.dim) --ns;
        if (ns >= 2) {
            const int steps = (int)((nk + ns - 1) / ns);
            ns = (nk + steps - 1) / steps;
        }
        if (ns >= 2) {
            GemmArgs p = g;
            p.bias = nullptr; p.act = ACT_NONE; p.res_mode = RES_NONE; p.res = nullptr; p.res2 = nullptr;
            p.C = ws_split_; p.out_dtype = DT_F32; p.ldc = w.n; p.map_mode = MAP_LINEAR;
            p.nsplit = (int)ns; p.split_steps = (nk + (int)ns - 1) / (int)ns; p.c_split_stride = (size_t)g.M * w.n;
            TRY(launch_gemm(p, img_dt_, st));
            TRY(launch_conv_reduce_pad(ws_split_, (int)ns, p.c_split_stride, w.b, res, res2, out, out_relu, B, Ho, Wo, w.n, relu_out, img_dt_, st));
            if (relu_written) *relu_written = out_relu != nullptr;
            return 0;
        }
    }
    if (out_relu && !strict_ && gemm_epilogue_is_pad16(g, img_dt_)) {       // also keep ReLU(out): the next unit's conv input
        g.C_relu = out_relu;
        if (relu_written) *relu_written = true;
    }
    return igemm(g, st);
}

// FeatureFusionBlock_custom.forward (lseg_blocks.py:337-358) for refinenet r (4..1)
int Engine::refine(int r, int B, hipStream_t st, bool stop_before_upsample) {
    const int l = r - 1, H = lh_[l], W = lw_[l], F = cfg.features;
    Refine& R = refine_[l];
    // A residual unit reads ReLU(x) as its conv input and x as its skip (lseg_blocks.py:270-288).  Where the producer of x ran on the
    // specialised epilogue it also left ReLU(x) (rnr_ / sumr_): the conv then runs without the per-fragment clamp in its K-loop.
    const uint16_t* rcu2_in;
    bool rcu2_relu = rn_relu_ok_[l];
    const uint16_t* rcu2_conv_in = rn_relu_ok_[l] ? rnr_[l] : rn_[l];
    if (R.has_u1) {
        // output = path_{r+1} + resConfUnit1(layer_r_rn)   (:345-347), RCU: lseg_blocks.py:265-288
        TRY(conv3x3(rn_relu_ok_[l] ? rnr_[l] : rn_[l], R.u1.c1, nullptr, nullptr, t1_[l], B, H, W, 1, rn_relu_ok_[l] ? 0 : 1, 1, st));
        TRY(conv3x3(t1_[l], R.u1.c2, rn_[l], path_[l + 1], sum_[l], B, H, W, 1, 0, 0, st, sumr_[l], &rcu2_relu));
        rcu2_in = sum_[l];
        rcu2_conv_in = rcu2_relu ? sumr_[l] : sum_[l];
    } else {
        rcu2_in = rn_[l];
    }
    TRY(conv3x3(rcu2_conv_in, R.u2.c1, nullptr, nullptr, t1_[l], B, H, W, 1, rcu2_relu ? 0 : 1, 1, st));
    TRY(conv3x3(t1_[l], R.u2.c2, rcu2_in, nullptr, t2_[l], B, H, W, 1, 0, 0, st));
    if (stop_before_upsample) return 0;         // the commuted head takes it from here
    if (strict_) TRY(launch_upsample2x_nhwc_split(t2_[l], pl(t2_[l]), up_[l], pl(up_[l]), B, H, W, F, st));
    else TRY(launch_upsample2x_nhwc(t2_[l], up_[l], B, H, W, F, img_dt_, st));          // :352-354
    GemmArgs g;
    gemm_args_init(g);                                                                 // out_conv :356
    g.A = up_[l]; g.W = R.out_conv.w; g.M = B * 4 * H * W; g.N = F; g.K = F; g.lda = F; g.ldw = F;
    g.bias = R.out_conv.b; g.C = path_[l]; g.out_dtype = img_dt_; g.ldc = F;
    if (l > 0) { g.map_mode = MAP_PADDED; g.ho = 2 * H; g.wo = 2 * W; }
    else g.map_mode = MAP_LINEAR;
    return igemm(g, st);
}

// timing events come from a free list filled outside the timed region (lseg_set_profiling / flush): no hipEventCreate per forward
hipEvent_t Engine::get_event() {
    hipEvent_t e;
    if (!ev_free_.empty()) { e = ev_free_.back(); ev_free_.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return nullptr;
    ev_pool_.push_back(e);
    return e;
}

int Engine::reserve_events(int n) {
    while ((int)ev_free_.size() < n) {
        hipEvent_t e;
        LSEG_HIP_TRY(hipEventCreate(&e));
        ev_free_.push_back(e);
    }
    return 0;
}

hipEvent_t Engine::prof_begin(int fam, hipStream_t st) {
    if (!((prof_mask >> fam) & 1u)) return nullptr;
    hipEvent_t e = get_event();
    if (e) (void)hipEventRecord(e, st);
    return e;
}

void Engine::prof_end(int fam, hipEvent_t e0, double flops, hipStream_t st) {
    if (!e0) return;
    hipEvent_t e1 = get_event();
    if (!e1) return;
    (void)hipEventRecord(e1, st);
    pf_[fam].ev.push_back({e0, e1});
    pf_[fam].slot.flops = flops;
}

// event pairs one forward records under the current mask (sizing of the pre-created pool)
int Engine::events_per_forward() const {
    int n = 0;
    const int per_block[PF_N] = {0, 1, 1, 1, 1, 1, 2, 0};
    for (int f = 0; f < PF_N; ++f)
        if ((prof_mask >> f) & 1u) n += (f == PF_FWD || f == PF_CORR) ? 1 : per_block[f] * cfg.depth;
    return n;
}

int Engine::flush_events() {
    for (int f = 0; f < PF_N; ++f) {
        for (auto& p : pf_[f].ev) {
            LSEG_HIP_TRY(hipEventSynchronize(p.second));
            float ms = 0.f;
            LSEG_HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
            pf_[f].slot.total_ms += ms; pf_[f].slot.launches += 1;
        }
        pf_[f].ev.clear();
    }
    for (auto e : ev_pool_) ev_free_.push_back(e);       // recycled, not destroyed
    ev_pool_.clear();
    return 0;
}

// Range check of the 16-bit image-tower activations of the forwards run so far (every buffer ALLOC16 made + the fp16 head map):
// out[0] = non-finite values (inf / NaN: an fp16 operand tower overflows at 65504), out[1] = finite values with |x| >= 2^15 (within a
// factor 2 of the fp16 limit), out[2] = largest finite |x| as fp32 bits, out[3] = elements scanned.  Synchronises `st`.
int Engine::check_range(unsigned long long* host_out4, hipStream_t st) {
    if (!inited_) return set_error(LSEG_ERR_STATE, "engine not initialised");
    LSEG_HIP_TRY(hipMemsetAsync(range_out_, 0, 4 * sizeof(unsigned long long), st));
    const int dt = img_dt_ == DT_BF16 ? DT_BF16 : DT_F16;
    for (auto& b : act16_) {
        const int d = b.first == g16pad_ ? DT_F16 : dt;
        TRY(launch_range16(b.first, b.second, d, range_out_, st));
    }
    LSEG_HIP_TRY(hipMemcpyAsync(host_out4, range_out_, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    LSEG_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int Engine::overflow_seen(bool reset) {
    if (ovf_pending_ && hipEventQuery(ev_ovf_) == hipSuccess) {
        ovf_pending_ = false;
        if (ovf_host_[0]) ovf_sticky_ = true;
    }
    const int r = ovf_sticky_ ? 1 : 0;
    // the DEVICE flag is sticky too: it is cleared only here, after a report -- every later forward keeps copying the 1 until someone has seen it
    if (reset && r) {
        ovf_sticky_ = false;
        (void)hipMemsetAsync(ovf_dev_, 0, sizeof(unsigned), nullptr);      // (rare path: after a reported overflow)
    }
    return r;
}

int Engine::get_profile(const char* family, double* ms, int64_t* launches, double* flops) {
    TRY(flush_events());
    static const char* names[PF_N] = {"forward", "mlp_fc1", "mlp_fc2", "attn_proj", "attn_qkv", "attention", "layernorm", "correlation"};
    ProfileSlot* s = nullptr;
    for (int f = 0; f < PF_N; ++f)
        if (!strcmp(family, names[f])) s = &pf_[f].slot;
    if (!s) return set_error(LSEG_ERR_INVALID, "unknown profile family '%s'", family);
    if (ms) *ms = s->total_ms;
    if (launches) *launches = s->launches;
    if (flops) *flops = s->flops;
    return 0;
}

// Split-K for the two GEMMs that add into the fp32 residual stream (attn.proj K = D, mlp.fc2 K = 4D; N = D) when the batch is small:
// at B = 1 their 901 x 1024 output is 64 tiles of 128 x 128 on 256 CUs, and the K = 4096 contraction runs 64 K-steps deep on a
// quarter of the chip.  The K range is cut into ns work items per tile (>= 4 K-steps each, ~2 resident workgroups per CU in total);
// each writes an fp32 partial slab (EPI_PART32), and the LayerNorm that follows anyway adds bias + slabs into x_ in a fixed order
// (launch_layernorm_reduce) -- no extra launch, no atomics, deterministic.  Returns ns (0: the GEMM was left as it is).
int Engine::split_residual(GemmArgs& g, int M, int N, int K) {
    static const int enabled = getenv("LSEG_SPLITK") ? atoi(getenv("LSEG_SPLITK")) : 1;      // tools: 0 switches it off
    if (strict_ || !enabled || (cfg.flags & 4) || (N % 256) != 0) return 0;
    // Tile and split factor together, from measured per-K-step costs of the two tile shapes (us per 64-deep K-step of one work item on
    // one CU: 256x256 2.07, 128x128 0.87, 0.81 with two work items resident; ~4 K-steps of fixed cost per item; tools/step_probe.py
    // sweeps): a CU works through ceil(items / CUs) items; the slabs cost the LayerNorm ns * M * N * 4 bytes at ~4 TB/s.
    const int nk = K / 64, cus = device_cu_count(device);
    const long t_mid = (long)((M + 127) / 128) * (N / 128), t_huge = (long)((M + 255) / 256) * (N / 256);
    double best = 1e30; int best_ns = 1, best_tile = 0;
    for (int tile = 2; tile <= 6; tile += 4) {
        for (int ns = 1; ns <= 8; ns *= 2) {
            if (ns > 1 && (nk / ns < 4 || (size_t)ns * M > ws_split_rows_)) break;
            const long items = (tile == 2 ? t_mid : t_huge) * ns;
            const long per_cu = (items + cus - 1) / cus;
            const double step_us = tile == 2 ? (per_cu >= 2 ? 0.81 : 0.87) : 2.07;
            const double cost = per_cu * ((double)nk / ns + 4.0) * step_us + (ns > 1 ? (double)ns * M * N * 4.0 / 4.0e6 : 0.0);
            if (cost < best * 0.97) { best = cost; best_ns = ns; best_tile = tile; }       // a split must buy >= 3 %
        }
    }
    if (best_ns < 2) return 0;                      // unsplit: leave the launch (and its tile choice) to the GEMM's own model
    const int steps = (nk + best_ns - 1) / best_ns;
    const int ns = (nk + steps - 1) / steps;
    if (ns < 2) return 0;
    g.bias = nullptr; g.res_mode = RES_NONE; g.res = nullptr;
    g.C = ws_split_; g.nsplit = ns; g.split_steps = steps; g.c_split_stride = (size_t)M * N; g.tile_hint = best_tile;
    return ns;
}

int Engine::forward(const float* x_in, int B, float* logits, uint8_t* argmax_out, hipStream_t st) {
    if (!finalized_) return set_error(LSEG_ERR_STATE, "parameters not finalised (call lseg_finalize_params)");
    if (K_ < 1) return set_error(LSEG_ERR_STATE, "no text tokens set (call lseg_set_text_tokens)");
    if (B < 1 || B > cfg.max_batch) return set_error(LSEG_ERR_INVALID, "B=%d outside [1, max_batch=%d]", B, cfg.max_batch);
    if (!x_in) return set_error(LSEG_ERR_INVALID, "x is NULL");
    LSEG_HIP_TRY(hipSetDevice(device));
    if (train_mode) {                    // net.train(): activations saved for lseg_backward, BatchNorm on batch statistics
        if (argmax_out) return set_error(LSEG_ERR_UNSUPPORTED, "argmax output is an inference feature (train mode is on)");
        return forward_train(x_in, B, logits, st);
    }
    if (eval_stale_) TRY(finalize(st));  // optimizer steps refresh only the train-mode packs: fold BatchNorm / the commuted head again
    const lseg_config& c = cfg;
    const int D = c.dim, H = c.heads, F = c.features, M = B * ntok_;
    last_B_ = B;
    low_pending_ = false;
    hipEvent_t fwd0 = prof_begin(PF_FWD, st);
    const double ntok2 = (double)ntok_ * ntok_;

    // ---- text tower (lseg_net.py:181-183): re-run every call unless caching is on.  Its ~90 small
    // kernels are latency-bound, so they run on a side stream next to the image tower; fork/join
    // events keep everything ordered with respect to the caller's stream.
    const bool run_text = !text_external_ && (!text_cache || !text_valid);
    if (run_text) {
        LSEG_HIP_TRY(hipEventRecord(ev_fork_, st));
        LSEG_HIP_TRY(hipStreamWaitEvent(text_stream_, ev_fork_, 0));
        TRY(encode_text(text_stream_));
        LSEG_HIP_TRY(hipEventRecord(ev_join_, text_stream_));
    }

    // ---- forward_flex (lseg_vit.py:166-201): patch embed + cls + pos -------------------------------------
    if (strict_) TRY(launch_im2col_split(x_in, patchA_, pl(patchA_), B, c.img_h, c.img_w, c.patch, st));
    else TRY(launch_im2col_patch(x_in, patchA_, B, c.img_h, c.img_w, c.patch, img_dt_, st));
    GemmArgs g;
    gemm_args_init(g);
    g.A = patchA_; g.W = patch_.w; g.M = B * np_; g.N = D; g.K = patch_.k; g.lda = patch_.k; g.ldw = patch_.k;
    g.bias = patch_.b; g.res_mode = RES_PERIODIC; g.res = pos_; g.res_dtype = DT_F32; g.ldr = D;
    g.C = x_; g.out_dtype = DT_F32; g.ldc = D; g.map_mode = MAP_PERIODIC; g.p_div = np_; g.p_mul = ntok_; g.p_off = 1;
    TRY(igemm(g, st));
    TRY(launch_cls_rows(cls_, pos_, x_, B, ntok_, D, st));

    // ---- 24 x timm Block; hooks feed readout/reassemble/layer_rn immediately ------------------------------
    int pend_ns = 0;                    // > 0: x_ still lacks `pend_bias` + the pend_ns split-K slabs in ws_split_ (see split_residual)
    const float* pend_bias = nullptr;
    for (int i = 0; i < c.depth; ++i) {
        VitBlock& b = blocks_[i];
        hipEvent_t pe = prof_begin(PF_LN, st);
        if (strict_) TRY(launch_ln_split(x_, b.g1, b.b1, ln_, pl(ln_), M, D, 1e-6f, st));
        else if (pend_ns) { TRY(launch_layernorm_reduce(x_, ws_split_, pend_ns, (size_t)M * D, pend_bias, b.g1, b.b1, ln_, img_dt_, M, D, 1e-6f, st)); pend_ns = 0; }
        else TRY(launch_layernorm(x_, DT_F32, b.g1, b.b1, ln_, img_dt_, M, D, 1e-6f, st));
        prof_end(PF_LN, pe, 0.0, st);
        gemm_args_init(g);
        g.A = ln_; g.W = b.qkv.w; g.M = M; g.N = 3 * D; g.K = D; g.lda = D; g.ldw = D;
        g.bias = b.qkv.b; g.out_dtype = img_dt_; g.map_mode = MAP_QKV;
        g.C = q_; g.Ck = k_; g.Cv = vt_; g.qkv_dim = D; g.qkv_ntok = ntok_; g.qkv_npad = npad_; g.qkv_heads = H;
        // the softmax scale (head_dim^-0.5 = 0.125 for 64) * log2(e) rides on q through the QKV epilogue's single rounding: the attention
        // kernel's matrix pipe then delivers exp2 arguments (attention.hip PRE)
        g.qkv_qscale = 0.125f * 1.4426950408889634f;
        // fp16 operands only: with bf16 the pre-scaled body moved the mask flips at configs[1] from 1.57 % to 2.64 % (max |dlogit| 0.204 ->
        // 0.241; fp16 0.264 % vs 0.274 %: unchanged) -- lease J, profiles/r04_attention_experiments.txt.  LSEG_ATTN_PRE=0 / 1 (tools) forces a body.
        static const int pre_env = getenv("LSEG_ATTN_PRE") ? atoi(getenv("LSEG_ATTN_PRE")) : -1;
        const bool prescaled = !strict_ && gemm_qkv_scales_q(g, img_dt_) && (pre_env < 0 ? img_dt_ == DT_F16 : pre_env != 0);
        if (!prescaled) g.qkv_qscale = 0.f;
        pe = prof_begin(PF_QKV, st);
        TRY(igemm(g, st));
        prof_end(PF_QKV, pe, 2.0 * M * 3.0 * D * D, st);
        pe = prof_begin(PF_ATTN, st);
        if (strict_) TRY(launch_attention_strict(q_, k_, vt_, att_, pl(q_), pl(vt_), pl(att_), B, H, ntok_, npad_, 0.125f, st));
        else TRY(launch_attention_ex(q_, k_, vt_, att_, nullptr, B, H, ntok_, npad_, img_dt_, 0, 0.125f, prescaled ? 1 : 0, st));
        prof_end(PF_ATTN, pe, 4.0 * B * ntok2 * D, st);          // QK^T + PV, SURVEY 8(d): 4 N^2 D per image and block
        gemm_args_init(g);
        g.A = att_; g.W = b.proj.w; g.M = M; g.N = D; g.K = D; g.lda = D; g.ldw = D;
        g.bias = b.proj.b; g.res_mode = RES_DEST; g.res = x_; g.res_dtype = DT_F32;
        g.C = x_; g.out_dtype = DT_F32; g.ldc = D; g.map_mode = MAP_LINEAR; g.rows_alloc = rows_alloc_;
        pend_ns = split_residual(g, M, D, D);                 // small batches: partial slabs, summed into x_ by the LayerNorm that follows
        pend_bias = b.proj.b;
        pe = prof_begin(PF_PROJ, st);
        TRY(igemm(g, st));
        prof_end(PF_PROJ, pe, 2.0 * M * (double)D * D, st);
        pe = prof_begin(PF_LN, st);
        if (strict_) TRY(launch_ln_split(x_, b.g2, b.b2, ln_, pl(ln_), M, D, 1e-6f, st));
        else if (pend_ns) { TRY(launch_layernorm_reduce(x_, ws_split_, pend_ns, (size_t)M * D, pend_bias, b.g2, b.b2, ln_, img_dt_, M, D, 1e-6f, st)); pend_ns = 0; }
        else TRY(launch_layernorm(x_, DT_F32, b.g2, b.b2, ln_, img_dt_, M, D, 1e-6f, st));
        prof_end(PF_LN, pe, 0.0, st);
        gemm_args_init(g);
        g.A = ln_; g.W = b.fc1.w; g.M = M; g.N = 4 * D; g.K = D; g.lda = D; g.ldw = D;
        g.bias = b.fc1.b; g.act = ACT_GELU; g.C = mlp_; g.out_dtype = img_dt_; g.ldc = 4 * D; g.map_mode = MAP_LINEAR;
        g.tag = 1;
        pe = prof_begin(PF_FC1, st);
        TRY(igemm(g, st));
        prof_end(PF_FC1, pe, 2.0 * M * 4.0 * D * D, st);
        gemm_args_init(g);
        g.A = mlp_; g.W = b.fc2.w; g.M = M; g.N = D; g.K = 4 * D; g.lda = 4 * D; g.ldw = 4 * D;
        g.bias = b.fc2.b; g.res_mode = RES_DEST; g.res = x_; g.res_dtype = DT_F32;
        g.C = x_; g.out_dtype = DT_F32; g.ldc = D; g.map_mode = MAP_LINEAR; g.rows_alloc = rows_alloc_;
        pend_ns = split_residual(g, M, D, 4 * D);
        pend_bias = b.fc2.b;
        pe = prof_begin(PF_FC2, st);
        TRY(igemm(g, st));
        prof_end(PF_FC2, pe, 2.0 * M * 4.0 * D * D, st);
        // the slabs are normally summed by the next block's LayerNorm; a hooked block (the readout reads x_) and the last one need x_ now
        bool hooked = i + 1 == c.depth;
        for (int l = 0; l < 4; ++l) hooked = hooked || c.hooks[l] == i;
        if (pend_ns && hooked) {
            TRY(launch_layernorm_reduce(x_, ws_split_, pend_ns, (size_t)M * D, pend_bias, nullptr, nullptr, ln_, img_dt_, M, D, 1e-6f, st));
            pend_ns = 0;
        }

        for (int l = 0; l < 4; ++l) {
            if (c.hooks[l] != i) continue;
            if (debug) {
                if (!acts_[l]) ALLOC(acts_[l], float, (size_t)c.max_batch * ntok_ * D);
                LSEG_HIP_TRY(hipMemcpyAsync(acts_[l], x_, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            const int C = cp_[l];                 // padded channel count (== reassemble_ch unless ViT-B/32 level 1)
            // ProjectReadout (lseg_vit.py:86-90)
            if (strict_) TRY(launch_readout_cat_split(x_, catA_, pl(catA_), B, ntok_, D, st));
            else TRY(launch_readout_cat(x_, catA_, B, ntok_, D, img_dt_, st));
            gemm_args_init(g);
            g.A = catA_; g.W = readout_[l].w; g.M = B * np_; g.N = D; g.K = 2 * D; g.lda = 2 * D; g.ldw = 2 * D;
            g.bias = readout_[l].b; g.act = ACT_GELU; g.C = ro_; g.out_dtype = img_dt_; g.ldc = D; g.map_mode = MAP_LINEAR;
            TRY(igemm(g, st));
            // act_postprocess[3]: 1x1 conv (token-major rows == NHWC pixels, the Transpose/Unflatten are free)
            gemm_args_init(g);
            g.A = ro_; g.W = r1x1_[l].w; g.M = B * np_; g.N = C; g.K = D; g.lda = D; g.ldw = D;
            g.bias = r1x1_[l].b; g.out_dtype = img_dt_; g.ldc = C;
            if (c.resample_kind[l] == LSEG_RS_CONVT) { g.C = r1_; g.map_mode = MAP_LINEAR; }
            else if (c.resample_kind[l] == LSEG_RS_IDENTITY) { g.C = L_[l]; g.map_mode = MAP_PADDED; g.ho = gh_; g.wo = gw_; }
            else { g.C = tmp_pad_; g.map_mode = MAP_PADDED; g.ho = gh_; g.wo = gw_; }
            TRY(igemm(g, st));
            if (c.resample_kind[l] == LSEG_RS_CONVT) {
                // ConvTranspose2d(k = s = stride) as a GEMM with a pixel-shuffle scatter epilogue
                const int s = c.resample_k[l];
                gemm_args_init(g);
                g.A = r1_; g.W = rsmp_[l].w; g.M = B * np_; g.N = s * s * C; g.K = C; g.lda = C; g.ldw = C;
                g.bias = rsmp_[l].b; g.bias_mod = C; g.C = L_[l]; g.out_dtype = img_dt_; g.ldc = C;
                g.map_mode = MAP_PIXSHUF; g.ho = gh_; g.wo = gw_; g.ps_s = s; g.ps_C = C;
                TRY(igemm(g, st));
            } else if (c.resample_kind[l] == LSEG_RS_CONV_S2) {
                TRY(conv3x3(tmp_pad_, rsmp_[l], nullptr, nullptr, L_[l], B, gh_, gw_, 2, 0, 0, st));
            }
            // scratch.layerN_rn (lseg_net.py:171-174)
            TRY(conv3x3(L_[l], layer_rn_[l], nullptr, nullptr, rn_[l], B, lh_[l], lw_[l], 1, 0, 0, st, rnr_[l], &rn_relu_ok_[l]));
        }
    }

    // ---- refinenet4..1 (lseg_net.py:176-179) ---------------------------------------------------------------
    // Fast path of the head (everything but the debug taps / the split-precision mode): the two 1x1 convs after refinenet1's
    // upsample -- out_conv and head1 -- commute with it and run as ONE GEMM at the quarter resolution (elementwise.hip "commuted head")
    const bool commuted = !debug && !strict_ && headc_.n == c.out_c;
    for (int r = 4; r >= 1; --r) TRY(refine(r, B, st, commuted && r == 1));

    if (run_text) LSEG_HIP_TRY(hipStreamWaitEvent(st, ev_join_, 0));     // join before the correlation

    // ---- head1 + normalise + correlation (lseg_net.py:185-196) ------------------------------------------------
    const int h1 = 2 * lh_[0], w1 = 2 * lw_[0], hw1 = h1 * w1, Mp = B * hw1;
    const float logit_scale = expf(logf(1.0f / 0.07f));                // lseg_net.py:141
    gemm_args_init(g);
    g.A = path_[0]; g.W = head1_.w; g.M = Mp; g.N = c.out_c; g.K = F; g.lda = F; g.ldw = F;
    g.bias = head1_.b;
    // ... and so does the correlation: t_k . up(g)_P = up(t_k . g)_P, with ||up(g)_P|| from the dot products of g's 2x2 cells
    // (elementwise.hip "commuted correlation"): the pixel x text GEMM runs on 4x fewer pixels and the 240x240x512 feature map never exists
    static const bool corr_full = getenv("LSEG_CORR_FULLRES") != nullptr;           // A/B switch (tools): the correlation at (2h, 2w)
    const bool corr_low = commuted && !corr_full && (lw_[0] % 2) == 0;
    if (corr_low && group_k > 0) {
        if (K_ != B * group_k)
            return set_error(LSEG_ERR_INVALID, "grouped labels: %d token rows != B=%d x %d labels per image", K_, B, group_k);
        if (c.arch_option != 0) return set_error(LSEG_ERR_UNSUPPORTED, "per-image label sets have no head blocks (lseg_net_zs.py:177-214)");
    }
    if (corr_low) {
        const int hp = lh_[0] + 2, wp = lw_[0] + 2;
        g.A = t2_[0]; g.W = headc_.w; g.bias = headc_.b; g.M = B * hp * wp;
        g.C = g16pad_; g.out_dtype = DT_F16; g.ldc = c.out_c; g.map_mode = MAP_LINEAR;
        TRY(igemm(g, st));
        // the dedicated kernel (corr.hip): T resident in LDS, g streamed once, label planes and cell dot products from the same registers;
        // per-image label sets (zero-shot), other widths and label sets that do not fit the LDS take the generic GEMM + pixel_gram pair
        static const bool corr_generic = getenv("LSEG_CORR_GENERIC") != nullptr;    // A/B switch (tools, tests): the round-4 pair
        const bool corr_fused = !corr_generic && group_k == 0 && corr_planes_supported(K_, c.out_c);
        // "correlation" family = the dedicated kernel only (label planes + cell dot products; algorithmic BYTES in the flops slot): the generic
        // pair is not bracketed, so no event is taken that would never be paired (the pool is sized per family and forward)
        hipEvent_t pc = corr_fused ? prof_begin(PF_CORR, st) : nullptr;
        if (corr_fused) TRY(launch_corr_planes(g16pad_, tnorm_, rpl_, gram_, B, K_, lh_[0], lw_[0], c.out_c, st));
        else TRY(launch_pixel_gram(g16pad_, gram_, B, lh_[0], lw_[0], c.out_c, st));
        if (corr_fused) prof_end(PF_CORR, pc, (double)B * hp * wp * c.out_c * 2.0 + (double)B * K_ * lh_[0] * lw_[0] * 4.0 + (double)B * lh_[0] * lw_[0] * 20.0 +
                                              (double)K_ * c.out_c * 2.0, st);
        TRY(launch_norm_scale_plane(gram_, nscale_, B, lh_[0], lw_[0], logit_scale, st, ovf_dev_));
        // the sentinel travels to pinned host memory behind the forward; Engine::overflow_seen reads it once the event has passed (no sync)
        (void)overflow_seen(false);                      // harvest the previous forward's copy before its event is recorded again
        LSEG_HIP_TRY(hipMemcpyAsync(ovf_host_, ovf_dev_, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        LSEG_HIP_TRY(hipEventRecord(ev_ovf_, st));
        ovf_pending_ = true;
        if (corr_fused) {
        } else if (group_k > 0) {
            // lseg_net_zs.py:198-208: image b against its own k text rows -- B small GEMMs [k, out_c] x [out_c, hp*wp]
            for (int b = 0; b < B; ++b) {
                gemm_args_init(g);
                g.A = tnorm_ + (size_t)b * group_k * c.out_c; g.W = g16pad_ + (size_t)b * hp * wp * c.out_c;
                g.M = group_k; g.N = hp * wp; g.K = c.out_c; g.lda = c.out_c; g.ldw = c.out_c;
                g.C = rpl_ + (size_t)b * group_k * hp * wp; g.out_dtype = DT_F32; g.map_mode = MAP_LABELPLANES; g.p_div = hp * wp;
                TRY(launch_gemm(g, DT_F16, st));
            }
        } else {
            gemm_args_init(g);
            g.A = tnorm_; g.W = g16pad_; g.M = K_; g.N = B * hp * wp; g.K = c.out_c; g.lda = c.out_c; g.ldw = c.out_c;
            g.C = rpl_; g.out_dtype = DT_F32; g.map_mode = MAP_LABELPLANES; g.p_div = hp * wp;
            TRY(launch_gemm(g, DT_F16, st));
        }
        // R -> logits.  When only the full-resolution logits are wanted the two x2 upsamples run as one pass and the (2h, 2w) logits
        // stay in LDS; masks, head blocks, the "lowres" tap and lseg_forward_stats need them in memory (made on demand below).
        const int kk = group_k > 0 ? group_k : K_;
        low_pending_ = true; low_planes_ = B * kk; low_k_ = kk;
        static const bool no_4x = getenv("LSEG_NO_UPSAMPLE4X") != nullptr;          // A/B switch (tools)
        if (logits && !argmax_out && c.arch_option == 0 && !no_4x) {
            TRY(launch_upsample4x_planes_scaled(rpl_, nscale_, logits, B * kk, kk, lh_[0], lw_[0], st));
            last_low_ = low_; last_kout_ = kk;
            prof_end(PF_FWD, fwd0, 0.0, st);
            return 0;
        }
        TRY(materialize_low(st));
    } else if (commuted) {
        // g = Wc t2 + bc on every row of the padded map (border rows are never read), fp32; then x2 bilinear + L2-norm + fp16 casts
        g.A = t2_[0]; g.W = headc_.w; g.bias = headc_.b; g.M = B * (lh_[0] + 2) * (lw_[0] + 2);
        g.C = gpad_; g.out_dtype = DT_F32; g.ldc = c.out_c; g.map_mode = MAP_LINEAR;
        TRY(igemm(g, st));
        TRY(launch_upsample_norm_f16(gpad_, a16_, B, lh_[0], lw_[0], c.out_c, logit_scale, st));
    } else if (c.out_c == 512 && !debug && !strict_) {
        // fused: head1 + fp32 L2-norm + the two fp16 roundings in one epilogue (rows are complete inside
        // a workgroup); the 118 MB/image fp32 feature map is never written
        g.C = a16_; g.out_dtype = DT_F16; g.ldc = c.out_c; g.map_mode = MAP_ROWNORM; g.rn_scale = logit_scale;
        TRY(igemm(g, st));
    } else {
        g.C = feat_; g.out_dtype = DT_F32; g.ldc = c.out_c; g.map_mode = MAP_LINEAR;
        TRY(igemm(g, st));
        TRY(launch_l2norm_scale_f16(feat_, a16_, Mp, c.out_c, logit_scale, st));
    }
    const int Kout = group_k > 0 ? group_k : K_;            // label planes per image
    if (corr_low) {
        // low_ already holds the (2h, 2w) logits
    } else if (group_k > 0) {
        // lseg_net_zs.py:198-208: image b against its own k text rows -- B small GEMMs [hw1,out_c] x [out_c,k]
        if (K_ != B * group_k)
            return set_error(LSEG_ERR_INVALID, "grouped labels: %d token rows != B=%d x %d labels per image", K_, B, group_k);
        if (c.arch_option != 0) return set_error(LSEG_ERR_UNSUPPORTED, "per-image label sets have no head blocks (lseg_net_zs.py:177-214)");
        for (int b = 0; b < B; ++b) {
            gemm_args_init(g);
            g.A = a16_ + (size_t)b * hw1 * c.out_c; g.W = tnorm_ + (size_t)b * group_k * c.out_c;
            g.M = hw1; g.N = group_k; g.K = c.out_c; g.lda = c.out_c; g.ldw = c.out_c;
            g.round_mid = 1; g.C = low_ + (size_t)b * group_k * hw1; g.out_dtype = DT_F32; g.map_mode = MAP_NCHW; g.p_div = hw1;
            TRY(launch_gemm(g, DT_F16, st));
        }
    } else {
    // the labels are the GEMM's rows and the pixels its columns: each lane ends up with 4 consecutive pixels of one label plane
    // (16-byte stores into [B, K, h*w]) and K = 150 pads to 160 rows, not to 256 columns
    gemm_args_init(g);
    g.A = tnorm_; g.W = a16_; g.M = K_; g.N = Mp; g.K = c.out_c; g.lda = c.out_c; g.ldw = c.out_c;
    g.round_mid = 1; g.C = low_; g.out_dtype = DT_F32; g.map_mode = MAP_LABELPLANES; g.p_div = hw1;
    TRY(launch_gemm(g, DT_F16, st));
    }
    float* low = low_;
    if (c.arch_option == 1 || c.arch_option == 2) {                    // lseg_net.py:198-201
        const int bott = c.arch_option == 1;
        float* src = low_; float* dst = low2_;           // low_ stays intact (the "lowres" tap)
        for (int d = 0; d < c.block_depth - 1; ++d) {
            TRY(launch_head_block(src, dst, hb_w_, hb_b_, B, Kout, h1, w1, bott, c.activation, 1, st));
            src = dst; dst = (dst == low2_) ? low3_ : low2_;
        }
        TRY(launch_head_block(src, dst, hb_w_, hb_b_, B, Kout, h1, w1, bott, c.activation, 0, st));
        low = dst;
    }
    // masks = argmax over the labels of the x2-upsampled logits (what every caller of the reference computes from the full tensor:
    // torch.max(pred, 1), lsegmentation_module.py:114-117), read through the bilinear on the fly: no 138 MB / image logits needed
    last_low_ = low; last_kout_ = Kout;
    if (argmax_out && Kout > 256) return set_error(LSEG_ERR_UNSUPPORTED, "uint8 masks need K <= 256 (K=%d)", Kout);
    if (argmax_out) TRY(launch_seg_stats_ex(low, nullptr, B, Kout, 4 * hw1, -1, nullptr, nullptr, argmax_out, 1, h1, w1, st));
    // ---- scratch.output_conv: bilinear x2, align_corners=True (lseg_net.py:203) ----------------------------------
    if (logits) TRY(launch_upsample2x_planes(low, logits, B * Kout, h1, w1, st));
    prof_end(PF_FWD, fwd0, 0.0, st);
    return 0;
}

// the (2h, 2w) logits of the last forward in memory (the one-pass x4 upsample skipped them)
int Engine::materialize_low(hipStream_t st) {
    if (!low_pending_) return 0;
    low_pending_ = false;
    return launch_upsample2x_planes_scaled(rpl_, nscale_, low_, low_planes_, low_k_, lh_[0], lw_[0], st);
}

// pixAcc / IoU counts and the cross-entropy sum of the LAST forward's output against a target mask, from the low-resolution logits
// through the x2 bilinear on the fly (lsegmentation_module.py:49-50,59-60,72: the metric / loss step after the path)
int Engine::forward_stats(const int64_t* target, int ignore_index, int64_t* counts, double* nll, hipStream_t st) {
    if (!last_low_ || last_B_ < 1) return set_error(LSEG_ERR_STATE, "no forward has run");
    if (!target || !counts || !nll) return set_error(LSEG_ERR_INVALID, "forward_stats: NULL pointer");
    LSEG_HIP_TRY(hipSetDevice(device));
    TRY(materialize_low(st));
    const int h1 = 2 * lh_[0], w1 = 2 * lw_[0];
    return launch_seg_stats_ex(last_low_, target, last_B_, last_kout_, 4 * h1 * w1, ignore_index, reinterpret_cast<unsigned long long*>(counts),
                               nll, nullptr, 1, h1, w1, st);
}

int Engine::get_text_features(void* out, hipStream_t st) {
    if (!text_valid) return set_error(LSEG_ERR_STATE, "text features not computed yet");
    LSEG_HIP_TRY(hipMemcpyAsync(out, tnorm_, (size_t)K_ * cfg.out_c * 2, hipMemcpyDeviceToDevice, st));
    return 0;
}

int Engine::get_intermediate(const char* name, float* out, size_t cap, size_t* n, hipStream_t st) {
    if (last_B_ < 1) return set_error(LSEG_ERR_STATE, "no forward has run");
    const int B = last_B_, F = cfg.features;
    size_t need_n = 0;
    if (!strncmp(name, "act", 3) && name[3] >= '1' && name[3] <= '4' && !name[4]) {
        const int l = name[3] - '1';
        if (!acts_[l]) return set_error(LSEG_ERR_STATE, "activation taps need lseg debug mode on before forward");
        need_n = (size_t)B * ntok_ * cfg.dim;
        if (cap < need_n) return set_error(LSEG_ERR_INVALID, "buffer too small: %zu < %zu", cap, need_n);
        LSEG_HIP_TRY(hipMemcpyAsync(out, acts_[l], need_n * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (!strncmp(name, "path", 4) && name[4] >= '1' && name[4] <= '4' && !name[5]) {
        const int l = name[4] - '1';
        const int H = 2 * lh_[l], W = 2 * lw_[l];
        need_n = (size_t)B * F * H * W;
        if (cap < need_n) return set_error(LSEG_ERR_INVALID, "buffer too small: %zu < %zu", cap, need_n);
        TRY(launch_nhwc_to_nchw_f32(path_[l], out, B, H, W, F, F, l > 0 ? 1 : 0, img_dt_, st, pl(path_[l])));
    } else if (!strncmp(name, "rn", 2) && name[2] >= '1' && name[2] <= '4' && !name[3]) {
        const int l = name[2] - '1';
        need_n = (size_t)B * F * lh_[l] * lw_[l];
        if (cap < need_n) return set_error(LSEG_ERR_INVALID, "buffer too small: %zu < %zu", cap, need_n);
        TRY(launch_nhwc_to_nchw_f32(rn_[l], out, B, lh_[l], lw_[l], F, F, 1, img_dt_, st, pl(rn_[l])));
    } else if (!strncmp(name, "layer", 5) && name[5] >= '1' && name[5] <= '4' && !name[6]) {
        const int l = name[5] - '1';
        need_n = (size_t)B * cfg.reassemble_ch[l] * lh_[l] * lw_[l];
        if (cap < need_n) return set_error(LSEG_ERR_INVALID, "buffer too small: %zu < %zu", cap, need_n);
        TRY(launch_nhwc_to_nchw_f32(L_[l], out, B, lh_[l], lw_[l], cfg.reassemble_ch[l], cp_[l], 1, img_dt_, st, pl(L_[l])));
    } else if (!strcmp(name, "image_features")) {
        const int hw1 = 4 * lh_[0] * lw_[0];
        need_n = (size_t)B * cfg.out_c * hw1;
        if (cap < need_n) return set_error(LSEG_ERR_INVALID, "buffer too small: %zu < %zu", cap, need_n);
        TRY(launch_rows_to_nchw_f32(feat_, out, B, hw1, cfg.out_c, st));
    } else if (!strcmp(name, "lowres")) {
        TRY(materialize_low(st));
        const int hw1 = 4 * lh_[0] * lw_[0];
        need_n = (size_t)B * (group_k > 0 ? group_k : K_) * hw1;
        if (cap < need_n) return set_error(LSEG_ERR_INVALID, "buffer too small: %zu < %zu", cap, need_n);
        LSEG_HIP_TRY(hipMemcpyAsync(out, low_, need_n * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        return set_error(LSEG_ERR_INVALID, "unknown intermediate '%s'", name);
    }
    if (n) *n = need_n;
    return 0;
}

}  // namespace lseg

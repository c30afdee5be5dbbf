// engine.h -- the whole-forward plan of LSeg on one MI355X (owned buffers, packed weights).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "ops.h"
#include "../../include/lseg_hip.h"

namespace lseg {

struct BoundParam {
    const void* ptr = nullptr;
    int dtype = 0;
    std::vector<int64_t> shape;
    size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

// w: packed [n, k] MFMA operand; b: fp32 bias.  Train mode adds wt = w transposed [k, n] (dgrad GEMM operand) for Linears and
// wd = the flipped / channel-swapped 3x3 weights (dgrad as a forward conv) for convs.
struct Lin { uint16_t* w = nullptr; float* b = nullptr; int n = 0, k = 0; uint16_t* wt = nullptr; uint16_t* wd = nullptr; };

struct VitBlock { float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; Lin qkv, proj, fc1, fc2; };
struct TextBlock { float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; Lin qkv, out, fc, proj; };
// c1/c2: BN-folded convs (eval).  Train mode: r1/r2 = the raw convs, BN affine parameters in fp32, per-unit saved maps.
struct Rcu {
    Lin c1, c2;
    Lin r1, r2;
    float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
    std::string key;                                                       // "scratch.refinenetR.resConfUnitU."
    uint16_t *cv1 = nullptr, *n1 = nullptr, *cv2 = nullptr;                // conv1 out, ReLU(bn1 out), conv2 out (padded NHWC)
    const uint16_t* in_relu = nullptr;                                     // ReLU(unit input) of the last train-mode forward (NULL: clamp in the K-loop)
    float *st1 = nullptr, *st2 = nullptr;                                  // [2C] batch sums of bn1 / bn2
};
struct Refine { Rcu u1, u2; Lin out_conv; bool has_u1 = false; };

struct ProfileSlot { double total_ms = 0; int64_t launches = 0; double flops = 0; };

// saved tensors of one ViT block for the backward (train mode): everything 16-bit is an MFMA operand of a wgrad / dgrad GEMM
struct BlockSave {
    float *xin = nullptr, *xmid = nullptr, *lse = nullptr;
    uint16_t *ln1 = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *att = nullptr, *ln2 = nullptr, *pre = nullptr, *mlp = nullptr;
};
struct LevelSave { uint16_t *cat = nullptr, *ropre = nullptr, *ro = nullptr, *r1 = nullptr, *tmp = nullptr, *dro = nullptr; };

typedef void (*lseg_reduce_fn)(void* user, void* dev_ptr, int64_t n_floats, void* stream);   // in-place sum over ranks
typedef void (*lseg_bucket_fn)(void* user, int bucket, void* stream);                        // the bucket's gradients are enqueued

class Engine {
public:
    Engine(const lseg_config& c, int device);
    ~Engine();
    int init();                               // allocate workspace
    int bind(const char* key, const void* p, int dtype, const int64_t* shape, int ndim);
    int finalize(hipStream_t st);
    int set_tokens(const int64_t* host_tok, int K, int ctx);
    int set_text_features(const void* dev_feat_f16, int K, hipStream_t st);
    bool text_external_ = false;   // the text features were handed in (lseg_set_text_features): forward never runs the text tower
    int encode_text(hipStream_t st);
    int forward(const float* x, int B, float* logits, uint8_t* argmax_out, hipStream_t st);
    int get_text_features(void* out_f16, hipStream_t st);
    int forward_stats(const int64_t* target, int ignore_index, int64_t* counts, double* nll, hipStream_t st);
    int get_intermediate(const char* name, float* out, size_t cap, size_t* n, hipStream_t st);
    int get_profile(const char* family, double* ms, int64_t* launches, double* flops);
    // 16-bit range check of the image tower (fp16 operands saturate at 65504): scans every 16-bit activation buffer of the plan
    int check_range(unsigned long long* host_out4, hipStream_t st);
    // Always-on overflow sentinel of the inference forward: norm_scale_plane_kernel raises a device flag when the head feature map carries an
    // inf / NaN (anything non-finite in the 16-bit tower ends up there); the flag is copied to pinned host memory behind every forward and
    // read WITHOUT synchronising: 1 as soon as a COMPLETED forward was flagged (sticky until reset), 0 otherwise.
    int overflow_seen(bool reset);
    // ---- training step (train.hip; modules/lsegmentation_module.py:66-81) ----
    int set_train(bool on);
    int bind_grad(const char* key, float* dev_ptr);
    int grad_ptr(const char* key, float** out, size_t* n);
    int backward(const float* dlogits, const int64_t* target, int ignore_index, int accumulate, double* dev_loss2, hipStream_t st,
                 const float* dev_grad_scale = nullptr);
    int train_loss(const int64_t* target, int ignore_index, double* dev_loss2, int64_t* dev_counts2, hipStream_t st);
    int sgd_momentum(const char* key, float** out, size_t* n);
    int sgd_mark_initialized(bool on) { sgd_first_ = !on; return 0; }
    int sgd_step(float lr_pretrained, float lr_scratch, float momentum, float weight_decay, hipStream_t st);
    int n_buckets() const { return cfg.depth + 1; }
    int bucket_of(const std::string& key) const;
    lseg_reduce_fn bn_sync_fn = nullptr; void* bn_sync_user = nullptr; int bn_world = 1;
    lseg_bucket_fn bucket_fn = nullptr; void* bucket_user = nullptr;
    bool train_mode = false;

    lseg_config cfg;
    int device;
    bool text_cache = false, text_valid = false, debug = false;
    int group_k = 0;              // > 0: per-image label sets of this size (LSegNetZS), see lseg_set_text_grouping
    std::string err;

private:
    void* dalloc(size_t bytes, bool zero = true);
    int need(const std::string& key, BoundParam& out, std::initializer_list<int64_t> shape);
    int pack_linear(const std::string& wkey, const std::string& bkey, int n, int k, int dt, Lin& out, hipStream_t st, bool image = false);
    int pack_f32(const std::string& key, size_t n, float*& out, hipStream_t st);
    int pack_conv3(const std::string& wkey, const std::string& bn_prefix, const std::string& bias_key, int co, int ci,
                   int cop, int cip, Lin& out, hipStream_t st);
    int conv3x3(const void* in, const Lin& w, const void* res, const void* res2, void* out, int B, int H, int W,
                int stride, int relu_in, int relu_out, hipStream_t st, void* out_relu = nullptr, bool* relu_written = nullptr);
    int refine(int r, int B, hipStream_t st, bool stop_before_upsample = false);
    int flush_events();
    int materialize_low(hipStream_t st);
    bool low_pending_ = false; int low_planes_ = 0, low_k_ = 0;      // low_ = scaled x2 upsample of rpl_ not yet written (one-pass x4 upsample ran)
    // train.hip
    int train_alloc();
    int finalize_train(hipStream_t st);
    int forward_train(const float* x, int B, float* logits, hipStream_t st);
    int rcu_train(const uint16_t* in, const uint16_t* in_relu, Rcu& U, const uint16_t* res2, uint16_t* out, uint16_t* out_relu, int B, int H, int W,
                  hipStream_t st);
    int rcu_backward(const uint16_t* dout, const uint16_t* in, Rcu& U, uint16_t* din, int lev, int B, int H, int W, int acc, hipStream_t st);
    int linear_gelu_saved(GemmArgs& g, uint16_t* pre, uint16_t* out, hipStream_t st);
    int lin_bwd(const uint16_t* dy, int M, int N, int K, const uint16_t* x, const uint16_t* wt, uint16_t* dx, float* dw, float* db,
                int acc, hipStream_t st, int dw_rows = -1, const uint16_t* dgelu_pre = nullptr);
    int conv_bwd(const uint16_t* dy_pad, const uint16_t* x_pad, int relu_x, const Lin& w, uint16_t* dx_pad, float* dw_dst, int B, int H,
                 int W, int Cin, int Cout, int Ci_real, int Co_real, int acc, hipStream_t st);
    int pick_split(int M, int N, int nk, GemmArgs& g);
    int block_backward(int i, int B, int acc, hipStream_t st);
    int readout_backward(int l, int B, int acc, hipStream_t st);
    int reassemble_backward(int l, int B, int acc, hipStream_t st);
    int refine_backward(int r, int B, int acc, hipStream_t st);
    float* grad(const std::string& key, size_t n);
    void bucket_done(int b, hipStream_t st) { if (bucket_fn) bucket_fn(bucket_user, b, (void*)st); }
    int bn_sync(float* p, int n, hipStream_t st) { if (bn_sync_fn && bn_world > 1) bn_sync_fn(bn_sync_user, p, n, (void*)st); return 0; }

    std::map<std::string, BoundParam> bound_;
    std::vector<void*> allocs_;
    std::vector<std::pair<const uint16_t*, size_t>> act16_;     // every 16-bit image-tower activation buffer (check_range)
    unsigned long long* range_out_ = nullptr;
    unsigned* ovf_dev_ = nullptr; unsigned* ovf_host_ = nullptr; hipEvent_t ev_ovf_ = nullptr; bool ovf_pending_ = false, ovf_sticky_ = false;
    bool finalized_ = false, inited_ = false;
    int last_B_ = 0, last_kout_ = 0;
    const float* last_low_ = nullptr;

    // derived geometry
    int gh_, gw_, np_, ntok_, npad_, img_dt_;
    int rows_alloc_ = 0;                        // allocated rows of x_ / ln_ / att_ / mlp_: max_batch * ntok rounded up to 256 (gemm_asm.hip)
    bool strict_ = false;                       // split-precision mode (lseg_config.image_dtype == LSEG_F16_SPLIT)
    std::map<const void*, size_t> plane_;       // 16-bit buffer -> element offset of its lo plane
    size_t pl(const void* p) const { auto it = plane_.find(p); return it == plane_.end() ? 0 : it->second; }
    int igemm(GemmArgs& g, hipStream_t st);     // image-tower GEMM: fills the split-precision planes when strict_
    int split_residual(GemmArgs& g, int M, int N, int K);
    float* ws_split_ = nullptr; size_t ws_split_rows_ = 0;
    float* pack_tmp_ = nullptr; size_t pack_tmp_n_ = 0;
    uint16_t* relu_tmp_ = nullptr;
    int pack_tmp(size_t n, hipStream_t st);
    int lh_[4], lw_[4];          // spatial size of reassembled level l (0..3)
    int cp_[4];                  // reassemble channels rounded up to 64 (ViT-B/32: 96 -> 128), extra channels are 0
    int tnpad_;

    // ---- packed parameters -------------------------------------------------------------------
    Lin patch_;
    float *cls_ = nullptr, *pos_raw_ = nullptr, *pos_ = nullptr;
    std::vector<VitBlock> blocks_;
    Lin readout_[4], r1x1_[4], rsmp_[4], layer_rn_[4];
    Refine refine_[4];           // index r-1
    Lin head1_;
    Lin headc_;                  // head1 o refinenet1.out_conv as ONE 1x1 conv (applied before the x2 upsample: engine.hip "commuted head")
    float *headc_w32_ = nullptr, *gpad_ = nullptr;
    uint16_t* g16pad_ = nullptr; float *rpl_ = nullptr, *gram_ = nullptr, *nscale_ = nullptr;     // commuted correlation (engine.hip)
    float *hb_w_ = nullptr, *hb_b_ = nullptr;
    float *tok_emb_ = nullptr, *tpos_ = nullptr, *tlnf_g_ = nullptr, *tlnf_b_ = nullptr;
    std::vector<TextBlock> tblocks_;
    Lin tproj_;

    // ---- workspace -----------------------------------------------------------------------------
    float* x_ = nullptr;
    uint16_t *ln_ = nullptr, *q_ = nullptr, *k_ = nullptr, *vt_ = nullptr, *att_ = nullptr, *mlp_ = nullptr;
    uint16_t *patchA_ = nullptr, *catA_ = nullptr, *ro_ = nullptr, *r1_ = nullptr, *tmp_pad_ = nullptr;
    uint16_t* L_[4] = {};         // reassembled maps, padded NHWC
    uint16_t* rn_[4] = {};        // layerN_rn outputs, padded NHWC
    uint16_t *rnr_[4] = {}, *sumr_[4] = {};   // ReLU(rn_) / ReLU(sum_) written by the producing conv's epilogue (refine())
    bool rn_relu_ok_[4] = {};
    uint16_t *t1_[4] = {}, *sum_[4] = {}, *t2_[4] = {};   // refinenet temporaries per level
    uint16_t* up_[4] = {};        // upsampled (plain) per level
    uint16_t* path_[4] = {};      // path_r outputs: r=4..2 padded at next level's size, r=1 plain
    float* feat_ = nullptr;
    uint16_t* a16_ = nullptr;
    float *low_ = nullptr, *low2_ = nullptr, *low3_ = nullptr;
    float* acts_[4] = {};
    // text
    int K_ = 0;
    int tl_ = 0;                  // text positions actually computed (<= text_ctx), see set_tokens
    int64_t* d_tok_ = nullptr;
    int* d_eot_ = nullptr;
    uint16_t *tx_ = nullptr, *tln_ = nullptr, *tq_ = nullptr, *tk_ = nullptr, *tvt_ = nullptr, *tatt_ = nullptr,
             *tmlp_ = nullptr, *tpool_ = nullptr, *tfeat_ = nullptr, *tnorm_ = nullptr;

    // ---- train mode (allocated by set_train(true)) ------------------------------------------------------------------------
    bool train_alloc_ = false, train_fwd_valid_ = false;
    const int64_t* loss_target_ = nullptr; int loss_ignore_ = 0;      // train_loss() already ran seg_stats for this forward and this target
    int train_B_ = 0;
    std::vector<BlockSave> sv_;
    float* xlast_ = nullptr;               // output of the last block (= xin of a virtual block `depth`)
    LevelSave lv_[4];
    uint16_t *dmapA_[4] = {}, *dmapB_[4] = {}, *dmapC_[4] = {}, *dmapD_[4] = {};   // gradient maps per pyramid level (padded NHWC, zero border)
    uint16_t *dpath_[4] = {};              // d path_{l+1}: aliases the level-(l-1) map it is produced in
    uint16_t *dpath0_ = nullptr;           // d path_1 rows [B*4*h*w, F]
    uint16_t *drn_[4] = {}, *dL_[4] = {};
    uint16_t *rowsA_ = nullptr, *rowsB_ = nullptr;  // row-major 16-bit temporaries (GEMM operand views of gradient maps)
    uint16_t *ddil_ = nullptr, *dtmp_ = nullptr;    // stride-2 reassemble conv: dilated dY, d(1x1 output)
    uint16_t *tn16_ = nullptr;
    uint16_t *ws_a_ = nullptr, *ws_b_ = nullptr;    // transposed operands of the wgrad GEMMs
    size_t ws_a_n_ = 0, ws_b_n_ = 0;
    float* ws_part_ = nullptr; size_t ws_part_n_ = 0;   // split-K partial results of the weight-gradient GEMMs
    float* ws_dw_ = nullptr; size_t ws_dw_n_ = 0;   // wgrad output in the engine's packed layout before the re-layout into the parameter's
    float *ws_stats_ = nullptr, *zeros_ = nullptr, *ws_ln_ = nullptr;
    float* ws_det_ = nullptr; size_t ws_det_n_ = 0;     // cfg.flags bit 3: partial rows of the deterministic column reductions
    float *gx_ = nullptr, *dpos_ = nullptr, *lse_px_ = nullptr;     // lse_px_: per-pixel log-sum-exp of the up-sampled logits [B, 2h, 2w]
    char* attn_ws_ = nullptr;              // per-layer scratch of the attention backward (transposed / head-major operand copies)
    uint16_t *g16_ = nullptr, *dmlp_ = nullptr, *dln_ = nullptr, *datt_ = nullptr, *dqkv_ = nullptr, *dtok_ = nullptr;
    bool g16_valid_ = false;                             // g16_ currently equals the 16-bit rounding of gx_ (written by the LayerNorm backward)
    uint16_t *drows_ = nullptr, *da_ = nullptr, *df_ = nullptr, *tnT_ = nullptr;
    unsigned long long* counts_ = nullptr; double* nll_ = nullptr;
    struct GradSlot { float* ptr = nullptr; size_t n = 0; bool bound = false; };
    std::map<std::string, GradSlot> grads_;
    bool sgd_first_ = true;
    // the optimizer step as table-driven launches: one multi-tensor SGD (which also refreshes the same-layout copies finalize() keeps of a
    // parameter: `direct_`), one multi-matrix transpose for the W^T copies, then only the packs with a real re-layout
    struct DirectDst { uint16_t* w16 = nullptr; float* w32 = nullptr; bool conflict = false; };
    std::map<std::string, DirectDst> direct_;
    void note_direct(const std::string& key, uint16_t* w16, float* w32);
    bool sgd_owns(const std::string& key) const { return partial_pack_ && sgd_keys_.count(key) != 0; }
    std::map<std::string, int> sgd_keys_;   // parameters the SGD table updates (and whose direct copies it writes)
    bool partial_pack_ = false;            // finalize() running inside sgd_step: skip what the optimizer kernel already wrote and the frozen text tower
    bool eval_stale_ = false;              // eval-only packs (BN-folded convs, commuted head) are behind the masters: refreshed by the next eval forward
    SgdSeg* sgd_table_ = nullptr; int sgd_nseg_ = 0; unsigned sgd_blocks_ = 0; bool sgd_dirty_ = true;
    float* mom_flat_ = nullptr; size_t mom_flat_n_ = 0;
    TransposeJob* wt_table_ = nullptr; int wt_n_ = 0; unsigned wt_blocks_ = 0;
    int build_sgd_table();
    // small buffers that are accumulated into with atomics, zeroed by ONE launch per pass once their list is known (recorded during the
    // first pass, which still uses one memset each): [0] the forward's BatchNorm batch sums, [1] the backward's bias gradients
    struct ZeroSet { std::vector<ZeroJob> host; ZeroJob* dev = nullptr; int n = 0; bool ready = false; };
    ZeroSet zero_fwd_, zero_bwd_;
    bool zero_note(ZeroSet& z, float* p, size_t n);       // true: the buffer was zeroed by this pass's launch_zero_multi
    int zero_begin(ZeroSet& z, hipStream_t st);
    int zero_end(ZeroSet& z);
    int bias_sum(const uint16_t* dy, float* db, int R, int C, int ld, int acc, hipStream_t st);

    // ---- side stream: the (small, latency-bound) text tower overlaps the image tower ----------------
    hipStream_t text_stream_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr, ev_text_done_ = nullptr;
    bool text_pending_ = false;

    // ---- profiling: HIP-event pairs around kernel families on the caller's stream (lseg_set_profiling mask bit = family index) --------
public:
    enum ProfFamily { PF_FWD = 0, PF_FC1, PF_FC2, PF_PROJ, PF_QKV, PF_ATTN, PF_LN, PF_CORR, PF_N };
    unsigned prof_mask = 0;
    int reserve_events(int n);
    int events_per_forward() const;
private:
    struct ProfFam { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; ProfileSlot slot; };
    ProfFam pf_[PF_N];
    std::vector<hipEvent_t> ev_pool_, ev_free_;
    hipEvent_t get_event();
    hipEvent_t prof_begin(int fam, hipStream_t st);
    void prof_end(int fam, hipEvent_t e0, double flops, hipStream_t st);
};

}  // namespace lseg

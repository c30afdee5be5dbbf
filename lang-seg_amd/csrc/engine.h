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

struct Lin { uint16_t* w = nullptr; float* b = nullptr; int n = 0, k = 0; };

struct VitBlock { float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; Lin qkv, proj, fc1, fc2; };
struct TextBlock { float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; Lin qkv, out, fc, proj; };
struct Rcu { Lin c1, c2; };
struct Refine { Rcu u1, u2; Lin out_conv; bool has_u1 = false; };

struct ProfileSlot { double total_ms = 0; int64_t launches = 0; double flops = 0; };

class Engine {
public:
    Engine(const lseg_config& c, int device);
    ~Engine();
    int init();                               // allocate workspace
    int bind(const char* key, const void* p, int dtype, const int64_t* shape, int ndim);
    int finalize(hipStream_t st);
    int set_tokens(const int64_t* host_tok, int K, int ctx);
    int encode_text(hipStream_t st);
    int forward(const float* x, int B, float* logits, uint8_t* argmax_out, hipStream_t st);
    int get_text_features(void* out_f16, hipStream_t st);
    int get_intermediate(const char* name, float* out, size_t cap, size_t* n, hipStream_t st);
    int get_profile(const char* family, double* ms, int64_t* launches, double* flops);

    lseg_config cfg;
    int device;
    bool text_cache = false, text_valid = false, profiling = false, debug = false;
    int group_k = 0;              // > 0: per-image label sets of this size (LSegNetZS), see lseg_set_text_grouping
    std::string err;

private:
    void* dalloc(size_t bytes, bool zero = true);
    int need(const std::string& key, BoundParam& out, std::initializer_list<int64_t> shape);
    int pack_linear(const std::string& wkey, const std::string& bkey, int n, int k, int dt, Lin& out, hipStream_t st);
    int pack_f32(const std::string& key, size_t n, float*& out, hipStream_t st);
    int pack_conv3(const std::string& wkey, const std::string& bn_prefix, const std::string& bias_key, int co, int ci,
                   int cop, int cip, Lin& out, hipStream_t st);
    int conv3x3(const void* in, const Lin& w, const void* res, const void* res2, void* out, int B, int H, int W,
                int stride, int relu_in, int relu_out, hipStream_t st);
    int refine(int r, int B, hipStream_t st);
    int flush_events();

    std::map<std::string, BoundParam> bound_;
    std::vector<void*> allocs_;
    bool finalized_ = false, inited_ = false;
    int last_B_ = 0;

    // derived geometry
    int gh_, gw_, np_, ntok_, npad_, img_dt_;
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

    // ---- side stream: the (small, latency-bound) text tower overlaps the image tower ----------------
    hipStream_t text_stream_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;

    // ---- profiling ---------------------------------------------------------------------------------
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_fc1_, ev_fwd_;
    std::vector<hipEvent_t> ev_pool_;
    ProfileSlot prof_fc1_, prof_fwd_;
    hipEvent_t get_event();
};

}  // namespace lseg

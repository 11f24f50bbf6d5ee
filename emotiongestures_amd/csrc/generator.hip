// Host-side plans (weight-arena manifests) and launch sequences of the whole generator / CVAE.
// Reference wiring: Transformer.forward, Full_model/Models_spatial_memory.py:566-616 (Models_memory.py:521-565);
// MLP_Reconstruct_v3.sample / forward, CAVE/BEAT_CVAE.py:403-447.
//
// One eg_generator_forward call enqueues every kernel of a batch on the caller's stream from C++ (no Python
// between launches), so it can be captured into a hipGraph by the host and replayed.
#include "common.h"
#include <string>
#include <vector>
#include <string.h>
#include <map>
#include <mutex>

// ---- error plumbing ----------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void eg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* eg_last_error(void) { return g_err; }
std::atomic<long long> g_eg_launches{0};
extern "C" int64_t eg_launch_count(void) { return (int64_t)g_eg_launches.load(std::memory_order_relaxed); }
// launches by label since the last reset (only counted under EG_LAUNCH_HIST=1): "label count\n" lines into buf, returns the bytes needed
bool g_eg_launch_hist = [] { const char* e = getenv("EG_LAUNCH_HIST"); return e && e[0] == '1'; }();
namespace {
std::mutex g_hist_mu;
std::map<std::string, long long> g_hist;
}  // namespace
void egi_count_launch(const char* what) {
    std::lock_guard<std::mutex> lk(g_hist_mu);
    ++g_hist[what];
}
extern "C" int64_t eg_launch_histogram(char* buf, int64_t cap, int32_t reset) {
    std::lock_guard<std::mutex> lk(g_hist_mu);
    std::string out;
    for (const auto& kv : g_hist) out += kv.first + " " + std::to_string(kv.second) + "\n";
    if (buf && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    if (reset) g_hist.clear();
    return (int64_t)out.size() + 1;
}
extern "C" const char* eg_version(void) { return "emogest-hip 0.1 (gfx950)"; }
static int g_default_precision = EG_PREC_F32;
extern "C" int eg_set_default_precision(int p) {
    EG_REQUIRE(p >= 0 && p <= 2, EG_ERR_BAD_ARG, "eg_set_default_precision: %d", p);
    g_default_precision = p;
    return EG_OK;
}
extern "C" int eg_get_default_precision(void) { return g_default_precision; }

// ---- per-device kernel attributes -----------------------------------------------------------------------------------
int eg_ensure_dynamic_lds(const void* kernel, size_t bytes, const char* who) {
    if (bytes <= 64 * 1024) return EG_OK;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;         // (kernel, device) -> largest size granted so far
    int dev = 0;
    EG_HIP_TRY(hipGetDevice(&dev), who);
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[std::make_pair(kernel, dev)];
    if (have >= bytes) return EG_OK;
    if (bytes > 160 * 1024) { eg_set_error("%s: needs %zu B of LDS (160 KiB per CU)", who, bytes); return EG_ERR_UNSUPPORTED; }
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        eg_set_error("%s: cannot reserve %zu B of LDS on device %d: %s", who, bytes, dev, hipGetErrorString(e));
        return EG_ERR_HIP;
    }
    have = bytes;
    return EG_OK;
}

// ---- launch profiler ---------------------------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int64_t tag; double flops; bool ok; int wgs; };
std::vector<ProfRec> g_prof;
int g_prof_n = 0;
bool g_prof_on = false;
}  // namespace
EgProfScope::EgProfScope(int64_t tag, double flops, hipStream_t s) : slot(-1), st(s) {
    if (!g_prof_on || g_prof_n >= (int)g_prof.size()) return;
    slot = g_prof_n++;
    g_prof[slot].tag = tag;
    g_prof[slot].flops = flops;
    g_prof[slot].wgs = 0;
    g_prof[slot].ok = hipEventRecord(g_prof[slot].a, st) == hipSuccess;
}
void EgProfScope::workgroups(int n) {
    if (slot >= 0) g_prof[slot].wgs = n;
}
EgProfScope::~EgProfScope() {
    if (slot >= 0 && hipEventRecord(g_prof[slot].b, st) != hipSuccess) g_prof[slot].ok = false;
}
extern "C" int eg_profile_enable(int32_t max_records) {
    EG_REQUIRE(max_records > 0, EG_ERR_BAD_ARG, "eg_profile_enable: max_records=%d", max_records);
    while ((int)g_prof.size() < max_records) {
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { eg_set_error("hipEventCreate failed"); return EG_ERR_HIP; }
        r.tag = 0; r.flops = 0; r.ok = false; r.wgs = 0;
        g_prof.push_back(r);
    }
    g_prof_n = 0;
    g_prof_on = true;
    return EG_OK;
}
extern "C" int eg_profile_disable(void) { g_prof_on = false; return EG_OK; }
extern "C" int32_t eg_profile_read_workgroups(int32_t* workgroups, int32_t capacity) {
    const int n = g_prof_n < capacity ? g_prof_n : capacity;
    for (int i = 0; i < n; ++i) workgroups[i] = g_prof[i].wgs;
    return n;
}
extern "C" int32_t eg_profile_read(int64_t* tags, double* flops, float* ms, int32_t capacity) {
    int n = g_prof_n < capacity ? g_prof_n : capacity;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        if (!g_prof[i].ok || hipEventSynchronize(g_prof[i].b) != hipSuccess || hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) {
            eg_set_error("eg_profile_read: event %d was not recorded / cannot be read", i);
            g_prof_n = 0;
            return EG_ERR_HIP;          // negative: no records
        }
        tags[i] = g_prof[i].tag; flops[i] = g_prof[i].flops; ms[i] = t;
    }
    g_prof_n = 0;
    return n;
}

// ---- internal launchers implemented in misc.hip ---------------------------------------------------------------
struct EgiPriorW {
    const float *w1, *b1, *s1, *t1, *w2, *b2, *s2, *t2;
    const float *sp_w0, *sp_b0, *sp_w1, *sp_b1, *tc_w0, *tc_b0, *tc_w1, *tc_b1, *tm_w0, *tm_b0, *tm_w1, *tm_b1;
};
int egi_embedding(const int64_t* idx, const float* table, float* out, int rows, int dim, int ld, int n_words, hipStream_t st);
int egi_add(const float* a, const float* b, float* out, size_t n, int row_len, int period, hipStream_t st);
int egi_time_linear(const float* x, const float* w, const float* bias, float* y, int batch, int L, int C, int ld, hipStream_t st);
int egi_copy2d(const float* src, int lds_, float* dst, int ldd, int rows, int cols, hipStream_t st);
int egi_linear(const EgiLinear& p, hipStream_t st);
int egi_layernorm(const float* x, const float* gamma, const float* beta, float* y, void* img, int rows, int d, float eps, hipStream_t st);
int egi_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, float* attn,
                  void* oimg, int batch, int heads, int lq, int lk, int dk, int precision, hipStream_t st);
int egi_add_bcast(const float* a, const float* b, float* out, size_t rows, int row_len, int period, int rep, hipStream_t st);
int egi_prior_encoder(const float* prior, const EgiPriorW& w, float* cat, float* tm_mem, float* tm_pe, float* tm_gram, int batch,
                      int P, int F, int D, int Dpad, int chunk, int variant, hipStream_t st);
int egi_conv1d(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int n, int cin,
               int cout, int lin, int k, int stride, int pad, int act, hipStream_t st);
int egi_convt1d(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int n, int cin,
                int cout, int lin, hipStream_t st);
int egi_small_linear(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int n, int in, int out,
                     hipStream_t st);

#define EG_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc) return _rc;      \
    } while (0)

namespace {

// ---- manifest builder ---------------------------------------------------------------------------------------
struct Layout {
    std::vector<EgWeightEntry> entries;
    int64_t total = 0;
    int64_t add(const std::string& key, int kind, int d0, int d1, int d2, int d3, int64_t numel) {
        total = eg_round_up(total, 16);
        EgWeightEntry e;
        memset(&e, 0, sizeof(e));
        snprintf(e.key, sizeof(e.key), "%s", key.c_str());
        e.kind = kind; e.dims[0] = d0; e.dims[1] = d1; e.dims[2] = d2; e.dims[3] = d3;
        e.offset = total; e.numel = numel;
        entries.push_back(e);
        total += numel;
        return e.offset;
    }
    int64_t raw(const std::string& key, int64_t numel) { return add(key, EG_PACK_RAW, (int)numel, 0, 0, 0, numel); }
    int64_t vec(const std::string& key, int n, int npad) { return add(key, EG_PACK_VEC_PAD, n, npad, 0, 0, npad); }
};

struct LinW { int64_t w = -1, b = -1; int n = 0, k = 0, npad = 0, kpad = 0; };
struct ConvW { int64_t w = -1, bias = -1, scale = -1, shift = -1; int cin = 0, cout = 0, coutp = 0, stride = 1; };
struct BlockW {
    ConvW c1, c2;
    int64_t se_w1, se_b1, se_w2, se_b2;
    bool ds = false;
    int64_t ds_w = -1, ds_scale = -1, ds_shift = -1;
    int cin, cout, stride;
};
struct MhaW { LinW q, kv, qkv, o; int64_t ln_g, ln_b; };
struct FfnW { LinW w1, w2; int64_t ln_g, ln_b; };

LinW add_linear(Layout& L, const std::string& prefix, int n, int k, bool bias, int kpad_force = 0, int npad_force = 0) {
    LinW w;
    w.n = n; w.k = k;
    w.npad = npad_force ? npad_force : (int)eg_round_up(n, 64);
    w.kpad = kpad_force ? kpad_force : (int)eg_round_up(k, 64);
    w.w = L.add(prefix + ".weight", EG_PACK_LINEAR, n, k, w.npad, w.kpad, (int64_t)w.npad * w.kpad * 2);
    if (bias) w.b = L.vec(prefix + ".bias", n, w.npad);
    return w;
}
ConvW add_conv3(Layout& L, const std::string& key, int cin, int cout, int stride) {
    ConvW c;
    c.cin = cin; c.cout = cout; c.coutp = (int)eg_round_up(cout, 16); c.stride = stride;
    c.w = L.add(key + ".weight", EG_PACK_CONV3X3, cout, cin, c.coutp, 0, (int64_t)9 * cin * c.coutp * 2);
    return c;
}
void add_bn(Layout& L, const std::string& prefix, int n, int npad, int64_t& scale, int64_t& shift) {
    scale = L.add(prefix, EG_PACK_BN_SCALE, n, npad, 0, 0, npad);
    shift = L.add(prefix, EG_PACK_BN_SHIFT, n, npad, 0, 0, npad);
}
// several nn.Linear weights sharing one input, concatenated along N into one packed image (keys joined by '|'):
// one launch with `count` times the workgroups instead of `count` small launches
LinW add_linear_cat(Layout& L, const std::vector<std::string>& prefixes, int n_each, int k, bool bias) {
    LinW w;
    const int n = n_each * (int)prefixes.size();
    w.n = n; w.k = k; w.npad = (int)eg_round_up(n, 64); w.kpad = (int)eg_round_up(k, 64);
    std::string wk, bk;
    for (size_t i = 0; i < prefixes.size(); ++i) {
        wk += (i ? "|" : "") + prefixes[i] + ".weight";
        bk += (i ? "|" : "") + prefixes[i] + ".bias";
    }
    w.w = L.add(wk, EG_PACK_LINEAR, n, k, w.npad, w.kpad, (int64_t)w.npad * w.kpad * 2);
    if (bias) w.b = L.vec(bk, n, w.npad);
    return w;
}
// a Dropout-only chain of nn.Linear layers (eval mode: one affine map) folded at pack time; `chains` joined along N by '|'
LinW add_linear_fold(Layout& L, const std::string& chains, int n, int k, int kpad_force = 0, int npad_force = 0) {
    LinW w;
    w.n = n; w.k = k;
    w.npad = npad_force ? npad_force : (int)eg_round_up(n, 64);
    w.kpad = kpad_force ? kpad_force : (int)eg_round_up(k, 64);
    w.w = L.add(chains, EG_PACK_LINEAR_FOLD, n, k, w.npad, w.kpad, (int64_t)w.npad * w.kpad * 2);
    w.b = L.add(chains, EG_PACK_BIAS_FOLD, n, w.npad, 0, 0, w.npad);
    return w;
}
// self_attn: fused Q|K|V projection of one input; cross attention: Q alone, K|V of the encoder output
MhaW add_mha(Layout& L, const std::string& p, int d, bool self_attn) {
    MhaW m;
    if (self_attn) {
        m.qkv = add_linear_cat(L, {p + ".w_qs", p + ".w_ks", p + ".w_vs"}, d, d, false);
    } else {
        m.q = add_linear(L, p + ".w_qs", d, d, false);
        m.kv = add_linear_cat(L, {p + ".w_ks", p + ".w_vs"}, d, d, false);
    }
    m.o = add_linear(L, p + ".fc", d, d, false);
    m.ln_g = L.raw(p + ".layer_norm.weight", d);
    m.ln_b = L.raw(p + ".layer_norm.bias", d);
    return m;
}
FfnW add_ffn(Layout& L, const std::string& p, int d, int di) {
    FfnW f;
    f.w1 = add_linear(L, p + ".w_1", di, d, true);
    f.w2 = add_linear(L, p + ".w_2", d, di, true);
    f.ln_g = L.raw(p + ".layer_norm.weight", d);
    f.ln_b = L.raw(p + ".layer_norm.bias", d);
    return f;
}

// ---- workspace carver ---------------------------------------------------------------------------------------
struct Carver {
    int64_t off = 0;    // bytes
    int64_t take(int64_t floats) {
        const int64_t o = off;
        off = eg_round_up(off + floats * (int64_t)sizeof(float), 256);
        return o;
    }
};

}  // namespace

struct EgGenerator {
    EgGeneratorConfig cfg;
    Layout L;
    // derived geometry
    int H1, W1, H2, W2, H3, W3, HW3, Dpad, stages[3] = {3, 4, 6}, filters[3] = {32, 64, 128};
    // weights
    int64_t stem_w, stem_b, stem_scale, stem_shift;
    std::vector<BlockW> blocks;
    ConvW final_conv;
    LinW a_fc1, a_fc2, emosem0, emo2, sem2, fus0, fus2, cls[4], post[4], prior_h0, prior_h2, txt_dec;
    bool fuse_se = true;        // identity SE blocks: gate from input moments + tail in conv2's epilogue (cfg.reserved[3] = 1 disables)
    bool fold = false;          // cfg.reserved[2]: the chains below replace their members
    bool shared_chip = false;   // cfg.reserved[4]: several batches in flight (ClipPipeline lanes): products choose their tile for CU time
    LinW f_audio, f_emo, f_sem, f_post, f_prior;
    int64_t pos_table;
    std::vector<MhaW> enc_attn, dec_attn;
    std::vector<FfnW> enc_ffn, dec_ffn;
    // prior encoder (raw layouts)
    int64_t pc_w1, pc_b1, pc_s1, pc_t1, pc_w2, pc_b2, pc_s2, pc_t2;
    int64_t sp_w0, sp_b0, sp_w1, sp_b1, tc_w0, tc_b0, tc_w1, tc_b1, tm_w0, tm_b0, tm_w1, tm_b1;
    // text
    int64_t emb;
    struct TcnConv { LinW tap0, tap1; int64_t bias; };
    std::vector<TcnConv> tcn;
    int64_t txt_fc1_w, txt_fc1_b;
    int Cpad;   // padded TCN channel stride
    bool keep_taps = false;
    // optional branch concurrency (cfg.reserved[1]): the text branch and the prior encoder are independent of the audio
    // tower (Models_spatial_memory.py:577-585), so they are forked onto two library-owned side streams and joined back
    // with events (capturable into a hipGraph as a fork/join).  Created lazily; no device memory involved.
    bool concurrent = false;
    mutable hipStream_t side[2] = {nullptr, nullptr};
    mutable hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    int ensure_streams() const {
        if (side[0]) return EG_OK;
        for (int i = 0; i < 2; ++i) {
            if (hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming) != hipSuccess) {
                eg_set_error("cannot create side stream/event");
                return EG_ERR_HIP;
            }
        }
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) { eg_set_error("cannot create event"); return EG_ERR_HIP; }
        return EG_OK;
    }
    ~EgGenerator() {
        for (int i = 0; i < 2; ++i) {
            if (side[i]) (void)hipStreamDestroy(side[i]);
            if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};

namespace {

struct GenWs {      // byte offsets into the workspace, for a given batch
    int64_t act[3], gap, gate, amap, afc1, afeat, emo_t, emo, sem_t, sem, cls_part, cls_h[3], cls_out;
    int64_t fus_in, fus_h, fusion, xa, xb, q, qkv, ao, proj, ffn_h, im_x[2], im_enc, im_h, im_a[3], im_p[2];
    int64_t prior_cat, prior_h, prior_enc, prior_rep, tm_mem, tm_pe, tm_gram, post_a, post_b, post_c, pose;
    int64_t t_emb, t_a, t_b, t_c, t_out;
    int64_t tap_stem, tap_l[3];
    int64_t total;
};

GenWs carve(const EgGenerator* g, int B, int NB = 0) {
    if (NB <= 0) NB = B;
    const EgGeneratorConfig& c = g->cfg;
    GenWs w;
    Carver cv;
    const int64_t BF = (int64_t)B * c.frames, D = c.d_model, NF = (int64_t)NB * c.frames;
    const int64_t act = (int64_t)B * g->H1 * g->W1 * 32;
    for (int i = 0; i < 3; ++i) w.act[i] = cv.take(act);
    {   // SE average-pool partials: conv2 of a block writes tiles * C_out floats per clip -- size for the largest block
        int64_t gap_per_clip = 0;
        int h = g->H1, wd = g->W1;
        for (const BlockW& b : g->blocks) {
            const int ho = (h + 2 - 3) / b.stride + 1, wo = (wd + 2 - 3) / b.stride + 1;
            const int64_t need = (int64_t)eg_conv3x3_gap_tiles(ho, wo, b.cout, b.cout, 1) * b.cout;
            gap_per_clip = need > gap_per_clip ? need : gap_per_clip;
            h = ho; wd = wo;
        }
        w.gap = cv.take((int64_t)B * gap_per_clip);
    }
    w.gate = cv.take((int64_t)B * 128);
    w.amap = cv.take(BF * g->HW3);
    w.afc1 = cv.take(BF * D); w.afeat = cv.take(BF * D);
    w.emo_t = cv.take(BF * D * 2); w.emo = cv.take(BF * D); w.sem_t = cv.take(16); w.sem = cv.take(BF * D);
    w.cls_part = cv.take((int64_t)(c.frames > 64 ? c.frames : 64) * B * D);
    w.cls_h[0] = cv.take((int64_t)B * D); w.cls_h[1] = cv.take((int64_t)B * 256); w.cls_h[2] = cv.take((int64_t)B * 64);
    w.cls_out = cv.take((int64_t)B * 16);
    w.fus_in = cv.take(NF * D); w.fus_h = cv.take(NF * D); w.fusion = cv.take(NF * D);
    w.xa = cv.take(NF * D); w.xb = cv.take(NF * D);
    w.q = cv.take(NF * D); w.qkv = cv.take(NF * D * 3); w.ao = cv.take(NF * D); w.proj = cv.take(NF * D);
    w.ffn_h = cv.take(NF * c.d_inner);
    {   // pre-split bf16 (hi, lo) images: 4 bytes per element, rows rounded up to the 64-row tile
        const int64_t N64 = eg_round_up(NF, 64), B64 = eg_round_up(BF, 64), KX = D > g->Dpad ? D : g->Dpad;
        w.im_x[0] = cv.take(N64 * KX); w.im_x[1] = cv.take(N64 * KX); w.im_enc = cv.take(N64 * D); w.im_h = cv.take(N64 * c.d_inner);
        w.im_a[0] = cv.take(B64 * D); w.im_a[1] = cv.take(B64 * D); w.im_a[2] = cv.take(B64 * D * 2);
        w.im_p[0] = cv.take(B64 * D); w.im_p[1] = cv.take(B64 * D);
    }
    w.prior_cat = cv.take(BF * g->Dpad); w.prior_h = cv.take(BF * D); w.prior_enc = cv.take(BF * D);
    w.prior_rep = cv.take(NB > B ? NF * D : 16);
    w.tm_mem = cv.take((int64_t)B * c.pose_dim); w.tm_pe = cv.take((int64_t)B * c.chunk + 16); w.tm_gram = cv.take((int64_t)c.pose_dim * c.chunk);
    w.post_a = cv.take(NF * D * 4); w.post_b = cv.take(NF * D); w.post_c = cv.take(NF * g->Dpad); w.pose = cv.take(NF * c.pose_dim);
    const int64_t BT = (int64_t)B * c.text_len;
    w.t_emb = cv.take(BT * g->Cpad); w.t_a = cv.take(BT * g->Cpad); w.t_b = cv.take(BT * g->Cpad); w.t_c = cv.take(BT * g->Cpad);
    w.t_out = cv.take(BT * 512);
    w.tap_stem = w.tap_l[0] = w.tap_l[1] = w.tap_l[2] = -1;
    if (g->keep_taps) {
        w.tap_stem = cv.take(act);
        w.tap_l[0] = cv.take(act);
        w.tap_l[1] = cv.take((int64_t)B * g->H2 * g->W2 * 64);
        w.tap_l[2] = cv.take((int64_t)B * g->H3 * g->W3 * 128);
    }
    w.total = cv.off;
    return w;
}

inline float* P(void* ws, int64_t off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + off); }

int run_linear(const float* arena, const LinW& w, const float* x, int lda, float* y, int ldc, int m, int relu, const float* res1,
               int ldr, int prec, hipStream_t st, int n_override = 0) {
    return eg_linear(x, lda, arena + w.w, w.kpad, w.b >= 0 ? arena + w.b : nullptr, res1, nullptr, ldr, y, ldc, m,
                     n_override ? n_override : w.n, w.kpad <= lda ? w.kpad : w.k, relu, 0, 0, prec, st);
}

int run_conv(const float* arena, const ConvW& c, const float* x, float* y, float* gap, int B, int h, int w, int relu, int nchw,
             int prec, hipStream_t st) {
    return eg_conv3x3(x, arena + c.w, c.bias >= 0 ? arena + c.bias : nullptr, c.scale >= 0 ? arena + c.scale : nullptr,
                      c.shift >= 0 ? arena + c.shift : nullptr, y, gap, B, h, w, c.cin, c.cout, c.stride, relu, nchw, prec, st);
}

// An activation as the products see it: fp32 rows and/or its bf16 (hi, lo) tile-planar images (bf16 modes only).
struct Act { float* f = nullptr; int ld = 0; void* img = nullptr; int kimg = 0; };
inline Act act(float* f, int ld, void* img = nullptr, int kimg = 0) { Act a; a.f = f; a.ld = ld; a.img = img; a.kimg = kimg; return a; }

// nn.Linear with optional pre-split input/output images.  In the bf16 modes an input that comes with images is consumed
// through them (no in-kernel split), and an output with an image slot is emitted split for the next product;
// want_f32 = false then skips the fp32 copy.  In f32 mode images are ignored and everything flows through fp32 buffers.
int lin(const EgGenerator* g, const float* arena, const LinW& w, const Act& x, int xk0, const Act& y, bool want_f32, int m, int relu,
        const float* res1, int ldr, hipStream_t st, int n_override = 0, int splits = 0, float* partial = nullptr) {
    const bool img_ok = g->cfg.precision != EG_PREC_F32;
    EgiLinear p;
    p.w = arena + w.w; p.ldw = w.kpad; p.bias = w.b >= 0 ? arena + w.b : nullptr;
    p.res1 = res1; p.ldr = ldr; p.m = m; p.n = n_override ? n_override : w.n; p.relu = relu; p.precision = g->cfg.precision;
    p.shared_chip = g->shared_chip ? 1 : 0;
    if (img_ok && x.img) {
        p.ximg = x.img; p.xK = x.kimg; p.xk0 = xk0; p.k = w.kpad;
    } else {
        p.x = x.f + xk0; p.lda = x.ld; p.k = (w.kpad <= x.ld - xk0 || w.kpad <= x.ld) && (x.ld >= w.kpad) ? w.kpad : w.k;
    }
    const bool yimg = img_ok && y.img;
    if (yimg) { p.yimg = y.img; p.yK = y.kimg; }
    if (want_f32 || !yimg) { p.y = y.f; p.ldc = y.ld; }
    if (splits > 1 && img_ok && x.img && !yimg) { p.splits = splits; p.partial = partial; }
    return egi_linear(p, st);
}

// MultiHeadAttention.forward (SubLayers.py:30-59).  xkv.f == nullptr: self attention (fused Q|K|V projection of xq).
// out receives LN(fc(attn) + xq) as fp32 and (bf16 modes) as images for the FFN that follows.
int run_mha(const EgGenerator* g, const float* arena, const MhaW& m, const Act& xq, const Act& xkv, const Act& out, const GenWs& w,
            void* ws, int B, int Lq, int Lk, hipStream_t st) {
    const int D = g->cfg.d_model;
    float *qkv = P(ws, w.qkv), *ao = P(ws, w.ao), *pr = P(ws, w.proj);
    void* aimg = g->cfg.precision != EG_PREC_F32 ? P(ws, w.im_h) : nullptr;       // attention output images (the FFN hidden slot is free here)
    if (!xkv.f) {
        EG_TRY(lin(g, arena, m.qkv, xq, 0, act(qkv, 3 * D), true, B * Lq, 0, nullptr, 0, st));
        EG_TRY(egi_attention(qkv, 3 * D, qkv + D, 3 * D, qkv + 2 * D, 3 * D, ao, D, nullptr, aimg, B, g->cfg.n_head, Lq, Lk, g->cfg.d_k, g->cfg.precision, st));
    } else {
        float* q = P(ws, w.q);
        EG_TRY(lin(g, arena, m.q, xq, 0, act(q, D), true, B * Lq, 0, nullptr, 0, st));
        EG_TRY(lin(g, arena, m.kv, xkv, 0, act(qkv, 2 * D), true, B * Lk, 0, nullptr, 0, st));
        EG_TRY(egi_attention(q, D, qkv, 2 * D, qkv + D, 2 * D, ao, D, nullptr, aimg, B, g->cfg.n_head, Lq, Lk, g->cfg.d_k, g->cfg.precision, st));
    }
    EG_TRY(lin(g, arena, m.o, act(ao, D, aimg, D), 0, act(pr, D), true, B * Lq, 0, xq.f, D, st));
    return egi_layernorm(pr, arena + m.ln_g, arena + m.ln_b, out.f, g->cfg.precision != EG_PREC_F32 ? out.img : nullptr, B * Lq, D, 1e-6f, st);
}

// PositionwiseFeedForward.forward (SubLayers.py:74-84); the 2048-wide hidden only exists as images in the bf16 modes
int run_ffn(const EgGenerator* g, const float* arena, const FfnW& f, const Act& x, const Act& out, const GenWs& w, void* ws, int rows,
            hipStream_t st) {
    const int D = g->cfg.d_model, DI = g->cfg.d_inner;
    float *h = P(ws, w.ffn_h), *pr = P(ws, w.proj);
    const Act hid = act(h, DI, P(ws, w.im_h), DI);
    EG_TRY(lin(g, arena, f.w1, x, 0, hid, false, rows, 1, nullptr, 0, st));
    // ONE clip (<= 64 rows: a single row tile) in a bf16 mode: w_2's 2048-deep product is 64 serial K steps on 8 workgroups; split K four ways (the
    // fp32 hidden buffer is free in these modes and takes the partials).  Only there: from two clips up every product accumulates K in one order
    // whatever the batch size, so a clip's pose does not depend on how a batch is chunked (nn.DataParallel's scatter, ClipPipeline's batches).
    const int sp = (g->cfg.precision != EG_PREC_F32 && rows <= 64 && DI >= 2048 && 4 * D <= DI) ? 4 : 0;
    EG_TRY(lin(g, arena, f.w2, hid, 0, act(pr, D), true, rows, 0, x.f, D, st, 0, sp, h));
    return egi_layernorm(pr, arena + f.ln_g, arena + f.ln_b, out.f, g->cfg.precision != EG_PREC_F32 ? out.img : nullptr, rows, D, 1e-6f, st);
}


// audio tower: spec [B,n_mels,T] -> audio_feat [B*F, D]  (ResNetSE34V2.py:62-74, Models_spatial_memory.py:118-133)
int run_audio_tower(const EgGenerator* g, const float* arena, const float* spec, const GenWs& w, void* ws, int B, hipStream_t st) {
    const int prec = g->cfg.precision, F = g->cfg.frames, D = g->cfg.d_model;
    float* bufs[3] = {P(ws, w.act[0]), P(ws, w.act[1]), P(ws, w.act[2])};
    float *gap = P(ws, w.gap), *gate = P(ws, w.gate);
    int xi = 0, h = g->H1, wd = g->W1, bi = 0;
    EG_TRY(eg_stem_conv(spec, arena + g->stem_w, arena + g->stem_b, arena + g->stem_scale, arena + g->stem_shift, bufs[0], B, g->H1,
                        g->W1, 32, st));
    if (g->keep_taps)
        EG_HIP_TRY(hipMemcpyAsync(P(ws, w.tap_stem), bufs[0], sizeof(float) * (size_t)B * g->H1 * g->W1 * 32, hipMemcpyDeviceToDevice, st), "tap copy");
    for (int s = 0; s < 3; ++s) {
        for (int j = 0; j < g->stages[s]; ++j, ++bi) {
            const BlockW& bw = g->blocks[bi];
            const int t1 = (xi + 1) % 3, t2 = (xi + 2) % 3;
            const int ho = (h + 2 - 3) / bw.stride + 1, wo = (wd + 2 - 3) / bw.stride + 1;
            if (!bw.ds && g->fuse_se) {
                // identity block: gate from the moments of conv1's output, then conv2 with relu(y * gate + x) in its epilogue --
                // y is never materialised and the separate tail pass disappears (conv.hip: se_gate_pre_kernel)
                EG_TRY(run_conv(arena, bw.c1, bufs[xi], bufs[t1], gap, B, h, wd, 1, 0, prec, st));
                const int tiles1 = eg_conv3x3_gap_tiles(h, wd, bw.cin, bw.cout, 1);
                EG_TRY(eg_se_gate_pre(bufs[t1], gap, tiles1, arena + bw.c2.w, arena + bw.c2.scale, arena + bw.c2.shift, arena + bw.se_w1,
                                      arena + bw.se_b1, arena + bw.se_w2, arena + bw.se_b2, gate, B, ho, wo, bw.cout, st));
                EG_TRY(eg_conv3x3_se(bufs[t1], arena + bw.c2.w, nullptr, arena + bw.c2.scale, arena + bw.c2.shift, gate, bufs[xi], bufs[t2], nullptr, B,
                                     ho, wo, bw.cout, bw.cout, 1, 0, 0, prec, st));
                xi = t2; h = ho; wd = wo;
                continue;
            }
            EG_TRY(run_conv(arena, bw.c1, bufs[xi], bufs[t1], nullptr, B, h, wd, 1, 0, prec, st));
            EG_TRY(run_conv(arena, bw.c2, bufs[t1], bufs[t2], gap, B, ho, wo, 0, 0, prec, st));
            const int tiles = eg_conv3x3_gap_tiles(ho, wo, bw.cout, bw.cout, 1);
            EG_TRY(eg_se_gate(gap, tiles, arena + bw.se_w1, arena + bw.se_b1, arena + bw.se_w2, arena + bw.se_b2, gate, B, bw.cout,
                              ho * wo, st));
            EG_TRY(eg_se_residual_relu(bufs[t2], gate, bufs[xi], bw.ds ? arena + bw.ds_w : nullptr, bw.ds ? arena + bw.ds_scale : nullptr,
                                       bw.ds ? arena + bw.ds_shift : nullptr, bufs[t1], B, ho, wo, bw.cout, h, wd, bw.cin, bw.stride, st));
            xi = t1; h = ho; wd = wo;
        }
        if (g->keep_taps)
            EG_HIP_TRY(hipMemcpyAsync(P(ws, w.tap_l[s]), bufs[xi], sizeof(float) * (size_t)B * h * wd * g->filters[s], hipMemcpyDeviceToDevice, st), "tap copy");
    }
    float* amap = P(ws, w.amap);
    EG_TRY(run_conv(arena, g->final_conv, bufs[xi], amap, nullptr, B, h, wd, 0, 1, prec, st));
    if (g->fold) return lin(g, arena, g->f_audio, act(amap, g->HW3), 0, act(P(ws, w.afeat), D, P(ws, w.im_a[1]), D), true, B * F, 0, nullptr, 0, st);
    const Act h1 = act(P(ws, w.afc1), D, P(ws, w.im_a[0]), D);
    EG_TRY(lin(g, arena, g->a_fc1, act(amap, g->HW3), 0, h1, false, B * F, 0, nullptr, 0, st));
    return lin(g, arena, g->a_fc2, h1, 0, act(P(ws, w.afeat), D, P(ws, w.im_a[1]), D), true, B * F, 0, nullptr, 0, st);
}

// text branch: TextEncoderTCN.forward (Models_spatial_memory.py:171-179) channels-last
int run_text(const EgGenerator* g, const float* arena, const int64_t* text, float* out, const GenWs& w, void* ws, int B, hipStream_t st) {
    const EgGeneratorConfig& c = g->cfg;
    const int L = c.text_len, C = c.tcn_hidden, ld = g->Cpad, rows = B * L, prec = c.precision;
    float *x = P(ws, w.t_emb), *h1 = P(ws, w.t_a), *h2 = P(ws, w.t_b), *y = P(ws, w.t_c);
    EG_TRY(egi_embedding(text, arena + g->emb, x, rows, c.embed_dim, ld, c.n_words, st));
    for (int i = 0; i < c.tcn_layers; ++i) {
        const int d = 1 << i;
        const EgGenerator::TcnConv& c1 = g->tcn[2 * i];
        const EgGenerator::TcnConv& c2 = g->tcn[2 * i + 1];
        // conv(x)[t] = b + W0 x[t-d] + W1 x[t]   (tcn.py:18-24: pad d, chomp d)
        EG_TRY(eg_linear(x, ld, arena + c1.tap0.w, c1.tap0.kpad, arena + c1.bias, nullptr, nullptr, 0, h1, ld, rows, C, C, 0, d, L, prec, st));
        EG_TRY(eg_linear(x, ld, arena + c1.tap1.w, c1.tap1.kpad, nullptr, h1, nullptr, ld, h1, ld, rows, C, C, 1, 0, 0, prec, st));
        EG_TRY(eg_linear(h1, ld, arena + c2.tap0.w, c2.tap0.kpad, arena + c2.bias, nullptr, nullptr, 0, h2, ld, rows, C, C, 0, d, L, prec, st));
        // relu(conv2) then relu(out + x)  (tcn.py:43-47)
        EG_TRY(eg_linear(h1, ld, arena + c2.tap1.w, c2.tap1.kpad, nullptr, h2, x, ld, y, ld, rows, C, C, 1, 0, 0, prec, st));
        float* t = x; x = y; y = t;
    }
    EG_TRY(egi_time_linear(x, arena + g->txt_fc1_w, arena + g->txt_fc1_b, h1, B, L, C, ld, st));
    return eg_linear(h1, ld, arena + g->txt_dec.w, g->txt_dec.kpad, arena + g->txt_dec.b, nullptr, nullptr, 0, out, 512, rows, 512, C, 0,
                     0, 0, prec, st);
}

int run_prior(const EgGenerator* g, const float* arena, const float* prior, const GenWs& w, void* ws, int B, hipStream_t st) {
    const EgGeneratorConfig& c = g->cfg;
    EgiPriorW pw;
    memset(&pw, 0, sizeof(pw));
    pw.w1 = arena + g->pc_w1; pw.b1 = arena + g->pc_b1; pw.s1 = arena + g->pc_s1; pw.t1 = arena + g->pc_t1;
    pw.w2 = arena + g->pc_w2; pw.b2 = arena + g->pc_b2; pw.s2 = arena + g->pc_s2; pw.t2 = arena + g->pc_t2;
    if (c.variant == 1) {
        pw.sp_w0 = arena + g->sp_w0; pw.sp_b0 = arena + g->sp_b0; pw.sp_w1 = arena + g->sp_w1; pw.sp_b1 = arena + g->sp_b1;
        pw.tc_w0 = arena + g->tc_w0; pw.tc_b0 = arena + g->tc_b0; pw.tc_w1 = arena + g->tc_w1; pw.tc_b1 = arena + g->tc_b1;
        pw.tm_w0 = arena + g->tm_w0; pw.tm_b0 = arena + g->tm_b0; pw.tm_w1 = arena + g->tm_w1; pw.tm_b1 = arena + g->tm_b1;
    }
    EG_TRY(egi_prior_encoder(prior, pw, P(ws, w.prior_cat), P(ws, w.tm_mem), P(ws, w.tm_pe), P(ws, w.tm_gram), B, c.prior_frames, c.frames,
                             c.pose_dim, g->Dpad, c.chunk, c.variant, st));
    const int rows = B * c.frames;
    if (g->fold)
        return lin(g, arena, g->f_prior, act(P(ws, w.prior_cat), g->Dpad), 0, act(P(ws, w.prior_enc), c.d_model, P(ws, w.im_p[1]), c.d_model), true, rows, 0,
                   nullptr, 0, st);
    const Act ph = act(P(ws, w.prior_h), c.d_model, P(ws, w.im_p[0]), c.d_model);
    EG_TRY(lin(g, arena, g->prior_h0, act(P(ws, w.prior_cat), g->Dpad), 0, ph, false, rows, 0, nullptr, 0, st));
    return lin(g, arena, g->prior_h2, ph, 0, act(P(ws, w.prior_enc), c.d_model, P(ws, w.im_p[1]), c.d_model), true, rows, 0, nullptr, 0, st);
}

// fusion -> encoder -> decoder -> post_projector for NB sequences (Models_spatial_memory.py:601-614)
int run_transformer(const EgGenerator* g, const float* arena, const float* fusion_in, const Act& prior_enc, float* pose, const GenWs& w,
                    void* ws, int NB, hipStream_t st) {
    const EgGeneratorConfig& c = g->cfg;
    const int F = c.frames, D = c.d_model, rows = NB * F;
    void *im0 = P(ws, w.im_x[0]), *im1 = P(ws, w.im_x[1]), *ime = P(ws, w.im_enc), *imh = P(ws, w.im_h);
    const Act fh = act(P(ws, w.fus_h), D, im0, D);
    EG_TRY(lin(g, arena, g->fus0, act(const_cast<float*>(fusion_in), D), 0, fh, false, rows, 1, nullptr, 0, st));
    EG_TRY(lin(g, arena, g->fus2, fh, 0, act(P(ws, w.fusion), D), true, rows, 0, nullptr, 0, st));
    float *xa = P(ws, w.xa), *xb = P(ws, w.xb);
    EG_TRY(egi_add(P(ws, w.fusion), arena + g->pos_table, xa, (size_t)rows * D, D, F, st));
    Act x = act(xa, D);                                  // first layer input has no images (comes from the positional add)
    for (int l = 0; l < c.n_layers; ++l) {
        const Act mid = act(xb, D, im0, D);
        const Act nxt = act(xa, D, l + 1 < c.n_layers ? im1 : ime, D);
        EG_TRY(run_mha(g, arena, g->enc_attn[l], x, Act(), mid, w, ws, NB, F, F, st));
        EG_TRY(run_ffn(g, arena, g->enc_ffn[l], mid, nxt, w, ws, rows, st));
        x = nxt;
    }
    const Act enc_out = x;                               // xa + images in im_enc, alive through the decoder
    float* da = P(ws, w.fus_h);                          // reuse (fusion hidden is dead)
    Act dx = prior_enc;
    for (int l = 0; l < c.n_layers; ++l) {
        const Act mid = act(xb, D, im0, D);
        const Act nxt = act(da, D, im1, D);
        EG_TRY(run_mha(g, arena, g->dec_attn[l], dx, enc_out, mid, w, ws, NB, F, F, st));
        EG_TRY(run_ffn(g, arena, g->dec_ffn[l], mid, nxt, w, ws, rows, st));
        dx = nxt;
    }
    if (g->fold) return lin(g, arena, g->f_post, dx, 0, act(pose, c.pose_dim), true, rows, 0, nullptr, 0, st);
    // post_projector: four affine layers chained through images (fp32 copies only in f32 mode)
    const Act pa = act(P(ws, w.post_a), D * 4, imh, D * 4), pb = act(P(ws, w.post_b), D, im0, D), pc = act(P(ws, w.post_c), g->Dpad, im1, g->Dpad);
    EG_TRY(lin(g, arena, g->post[0], dx, 0, pa, false, rows, 0, nullptr, 0, st));
    EG_TRY(lin(g, arena, g->post[1], pa, 0, pb, false, rows, 0, nullptr, 0, st));
    EG_TRY(lin(g, arena, g->post[2], pb, 0, pc, false, rows, 0, nullptr, 0, st, g->Dpad));
    return lin(g, arena, g->post[3], pc, 0, act(pose, c.pose_dim), true, rows, 0, nullptr, 0, st);
}


int validate_cfg(const EgGeneratorConfig& c) {
    EG_REQUIRE(c.frames > c.prior_frames && c.prior_frames >= c.chunk && c.chunk > 0, EG_ERR_BAD_ARG, "config: frames/prior/chunk inconsistent");
    EG_REQUIRE(c.d_model % 64 == 0 && c.d_model >= 64 && c.d_model <= 2048, EG_ERR_UNSUPPORTED, "config: d_model=%d", c.d_model);
    EG_REQUIRE(c.d_k == 64 && c.n_head * c.d_k == c.d_model, EG_ERR_UNSUPPORTED, "config: heads*d_k must equal d_model with d_k=64");
    EG_REQUIRE(c.n_mels % 16 == 0 && c.n_mels <= 512 && c.spec_len >= 4, EG_ERR_UNSUPPORTED, "config: spectrogram %dx%d", c.n_mels, c.spec_len);
    EG_REQUIRE(c.frames <= 128, EG_ERR_UNSUPPORTED, "config: frames=%d > 128 (final conv width)", c.frames);
    EG_REQUIRE(c.embed_dim == c.tcn_hidden && c.embed_dim % 4 == 0, EG_ERR_UNSUPPORTED, "config: TCN needs embed_dim == hidden, %%4");
    EG_REQUIRE(c.variant == 0 || c.variant == 1, EG_ERR_BAD_ARG, "config: variant=%d", c.variant);
    EG_REQUIRE(c.precision >= 0 && c.precision <= 2, EG_ERR_BAD_ARG, "config: precision=%d", c.precision);
    EG_REQUIRE(c.n_position >= c.frames, EG_ERR_BAD_ARG, "config: n_position < frames");
    EG_REQUIRE(c.pose_dim > 0 && c.pose_dim <= 1024 && c.n_words > 0 && c.text_len > 0 && c.text_len <= 128, EG_ERR_BAD_ARG, "config: sizes");
    return EG_OK;
}

}  // namespace

extern "C" int eg_generator_default_config(EgGeneratorConfig* c) {
    EG_REQUIRE(c, EG_ERR_BAD_ARG, "null config");
    memset(c, 0, sizeof(*c));
    c->frames = 34; c->pose_dim = 126; c->prior_frames = 4; c->chunk = 4; c->d_model = 512; c->d_inner = 2048; c->n_layers = 3;
    c->n_head = 8; c->d_k = 64; c->n_mels = 128; c->spec_len = 124; c->text_len = 60; c->n_words = 200; c->embed_dim = 300;
    c->tcn_hidden = 300; c->tcn_layers = 3; c->variant = 0; c->precision = g_default_precision; c->n_position = 60;
    return EG_OK;
}

extern "C" int eg_generator_create(const EgGeneratorConfig* cfg, EgGenerator** out) {
    EG_REQUIRE(cfg && out, EG_ERR_BAD_ARG, "eg_generator_create: null pointer");
    EG_TRY(validate_cfg(*cfg));
    EgGenerator* g = new EgGenerator();
    g->cfg = *cfg;
    g->keep_taps = cfg->reserved[0] != 0;
    g->concurrent = cfg->reserved[1] != 0;
    g->fold = cfg->reserved[2] != 0;
    g->fuse_se = cfg->reserved[3] == 0;
    g->shared_chip = cfg->reserved[4] != 0;
    const EgGeneratorConfig& c = g->cfg;
    g->H1 = c.n_mels; g->W1 = c.spec_len;
    g->H2 = (g->H1 - 1) / 2 + 1; g->W2 = (g->W1 - 1) / 2 + 1;
    g->H3 = (g->H2 - 1) / 2 + 1; g->W3 = (g->W2 - 1) / 2 + 1;
    g->HW3 = g->H3 * g->W3;
    if (g->HW3 % 8 != 0) { delete g; eg_set_error("config: final map %dx%d not a multiple of 8", g->H3, g->W3); return EG_ERR_UNSUPPORTED; }
    g->Dpad = (int)eg_round_up(c.pose_dim, 64);
    g->Cpad = (int)eg_round_up(c.tcn_hidden, 64);
    Layout& L = g->L;
    const int D = c.d_model, F = c.frames, P_ = c.prior_frames, PL = F - P_, PD = c.pose_dim;
    // --- audio tower (keys: audio_encoder.feat_extractor.*)
    const std::string fe = "audio_encoder.feat_extractor";
    g->stem_w = L.add(fe + ".conv1.weight", EG_PACK_STEM, 32, 0, 0, 0, 9 * 32);
    g->stem_b = L.raw(fe + ".conv1.bias", 32);
    add_bn(L, fe + ".bn1", 32, 32, g->stem_scale, g->stem_shift);
    int inpl = 32;
    for (int s = 0; s < 3; ++s) {
        const int planes = g->filters[s];
        for (int j = 0; j < g->stages[s]; ++j) {
            BlockW b;
            const std::string p = fe + ".layer" + std::to_string(s + 1) + "." + std::to_string(j);
            b.stride = (s > 0 && j == 0) ? 2 : 1;
            b.cin = inpl; b.cout = planes;
            b.c1 = add_conv3(L, p + ".conv1", inpl, planes, b.stride);
            add_bn(L, p + ".bn1", planes, planes, b.c1.scale, b.c1.shift);
            b.c2 = add_conv3(L, p + ".conv2", planes, planes, 1);
            add_bn(L, p + ".bn2", planes, planes, b.c2.scale, b.c2.shift);
            b.se_w1 = L.raw(p + ".se.fc.0.weight", planes / 8 * planes);
            b.se_b1 = L.raw(p + ".se.fc.0.bias", planes / 8);
            b.se_w2 = L.raw(p + ".se.fc.2.weight", planes * (planes / 8));
            b.se_b2 = L.raw(p + ".se.fc.2.bias", planes);
            b.ds = (b.stride != 1 || inpl != planes);
            if (b.ds) {
                b.ds_w = L.add(p + ".downsample.0.weight", EG_PACK_CONV1X1, planes, inpl, 0, 0, (int64_t)inpl * planes);
                add_bn(L, p + ".downsample.1", planes, planes, b.ds_scale, b.ds_shift);
            }
            g->blocks.push_back(b);
            inpl = planes;
        }
    }
    g->final_conv = add_conv3(L, "audio_encoder.final_conv1", 128, F, 1);
    if (g->final_conv.coutp > 64) {         // 65..128 frames run on the 128-wide kernel: pad the packed weights to 128 channels
        g->final_conv.coutp = 128;
        L.entries.back().dims[2] = 128;
        L.total -= L.entries.back().numel;
        L.entries.back().numel = (int64_t)9 * 128 * 128 * 2;
        L.total += L.entries.back().numel;
    }
    g->final_conv.bias = L.vec("audio_encoder.final_conv1.bias", F, g->final_conv.coutp);
    add_bn(L, "audio_encoder.bn1", F, g->final_conv.coutp, g->final_conv.scale, g->final_conv.shift);
    if (g->fold) {
        g->f_audio = add_linear_fold(L, "audio_encoder.fc2@audio_encoder.fc1", D, g->HW3);
        g->f_emo = add_linear_fold(L, "emotion_proj.2@emotion_proj.0", D, D);
        g->f_sem = add_linear_fold(L, "semantic_proj.2@semantic_proj.0", D, D);
    } else {
        g->a_fc1 = add_linear(L, "audio_encoder.fc1", D, g->HW3, true);
        g->a_fc2 = add_linear(L, "audio_encoder.fc2", D, D, true);
        // --- projections
        g->emosem0 = add_linear_cat(L, {"emotion_proj.0", "semantic_proj.0"}, D, D, true);      // both read spectrum_feature (:588-589)
        g->emo2 = add_linear(L, "emotion_proj.2", D, D, true);
        g->sem2 = add_linear(L, "semantic_proj.2", D, D, true);
    }
    g->fus0 = add_linear(L, "fusion_proj.0", D, D, true); g->fus2 = add_linear(L, "fusion_proj.2", D, D, true);
    g->cls[0] = add_linear(L, "emotion_classifer_header.0", D, F * D, true);
    g->cls[1] = add_linear(L, "emotion_classifer_header.2", 256, D, true);
    g->cls[2] = add_linear(L, "emotion_classifer_header.4", 64, 256, true);
    g->cls[3] = add_linear(L, "emotion_classifer_header.6", 8, 64, true);
    const std::string pp = "prior_seq_encoder";
    if (g->fold) {
        g->f_post = add_linear_fold(L, "post_projector.6@post_projector.4@post_projector.2@post_projector.0", PD, D);
        g->f_prior = add_linear_fold(L, pp + ".post_header.2@" + pp + ".post_header.0", D, PD, g->Dpad, 0);
    } else {
        g->post[0] = add_linear(L, "post_projector.0", D * 4, D, true);
        g->post[1] = add_linear(L, "post_projector.2", D, D * 4, true);
        g->post[2] = add_linear(L, "post_projector.4", PD, D, true, 0, g->Dpad);
        g->post[3] = add_linear(L, "post_projector.6", PD, PD, true, g->Dpad, 0);
        // --- prior encoder
        g->prior_h0 = add_linear(L, pp + ".post_header.0", D, PD, true, g->Dpad, 0);
        g->prior_h2 = add_linear(L, pp + ".post_header.2", D, D, true);
    }
    g->pc_w1 = L.raw(pp + ".pred_conv.0.weight", (int64_t)PL * P_ * 3); g->pc_b1 = L.raw(pp + ".pred_conv.0.bias", PL);
    add_bn(L, pp + ".pred_conv.2", PL, PL, g->pc_s1, g->pc_t1);
    g->pc_w2 = L.raw(pp + ".pred_conv.3.weight", (int64_t)PL * PL * 3); g->pc_b2 = L.raw(pp + ".pred_conv.3.bias", PL);
    add_bn(L, pp + ".pred_conv.5", PL, PL, g->pc_s2, g->pc_t2);
    if (c.variant == 1) {
        const int CD = c.chunk * PD;
        const std::string sp = pp + ".spatial_memory.spatial_chunk_encoder", tm = pp + ".temporal_memory";
        g->sp_w0 = L.raw(sp + ".0.weight", (int64_t)PD * CD); g->sp_b0 = L.raw(sp + ".0.bias", PD);
        g->sp_w1 = L.raw(sp + ".2.weight", (int64_t)PD * PD); g->sp_b1 = L.raw(sp + ".2.bias", PD);
        g->tc_w0 = L.raw(tm + ".temporal_chunk_encoder.0.weight", (int64_t)PD * CD); g->tc_b0 = L.raw(tm + ".temporal_chunk_encoder.0.bias", PD);
        g->tc_w1 = L.raw(tm + ".temporal_chunk_encoder.2.weight", (int64_t)PD * PD); g->tc_b1 = L.raw(tm + ".temporal_chunk_encoder.2.bias", PD);
        g->tm_w0 = L.raw(tm + ".temporal_memory_encoder.0.weight", (int64_t)c.chunk * CD); g->tm_b0 = L.raw(tm + ".temporal_memory_encoder.0.bias", c.chunk);
        g->tm_w1 = L.raw(tm + ".temporal_memory_encoder.2.weight", (int64_t)c.chunk * c.chunk); g->tm_b1 = L.raw(tm + ".temporal_memory_encoder.2.bias", c.chunk);
    }
    // --- transformer
    g->pos_table = L.add("encoder.position_enc.pos_table", EG_PACK_POS_TABLE, F, D, 0, 0, (int64_t)F * D);
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string e = "encoder.layer_stack." + std::to_string(l), d = "decoder.layer_stack." + std::to_string(l);
        g->enc_attn.push_back(add_mha(L, e + ".slf_attn", D, true));
        g->enc_ffn.push_back(add_ffn(L, e + ".pos_ffn", D, c.d_inner));
        g->dec_attn.push_back(add_mha(L, d + ".enc_attn", D, false));     // slf_attn parameters exist but are unused (Layers.py:52-53)
        g->dec_ffn.push_back(add_ffn(L, d + ".pos_ffn", D, c.d_inner));
    }
    // --- text branch
    g->emb = L.raw("text_encoder.embedding.weight", (int64_t)c.n_words * c.embed_dim);
    for (int i = 0; i < c.tcn_layers; ++i)
        for (int j = 1; j <= 2; ++j) {
            EgGenerator::TcnConv tc;
            const std::string p = "text_encoder.tcn.network." + std::to_string(i) + ".conv" + std::to_string(j);
            const int C = c.tcn_hidden, npad = (int)eg_round_up(C, 64);
            for (int tap = 0; tap < 2; ++tap) {
                LinW w;
                w.n = C; w.k = C; w.npad = npad; w.kpad = g->Cpad;
                w.w = L.add(p, EG_PACK_WN_TAP, C, C, tap, g->Cpad, (int64_t)npad * g->Cpad * 2);
                (tap == 0 ? tc.tap0 : tc.tap1) = w;
            }
            tc.bias = L.vec(p + ".bias", C, npad);
            g->tcn.push_back(tc);
        }
    g->txt_fc1_w = L.raw("text_encoder.fc1.0.weight", (int64_t)c.text_len * c.text_len);
    g->txt_fc1_b = L.raw("text_encoder.fc1.0.bias", c.text_len);
    g->txt_dec = add_linear(L, "text_encoder.decoder", 512, c.tcn_hidden, true, g->Cpad, 0);
    L.total = eg_round_up(L.total, 16);
    *out = g;
    return EG_OK;
}

extern "C" void eg_generator_destroy(EgGenerator* g) { delete g; }
extern "C" int64_t eg_generator_arena_floats(const EgGenerator* g) { return g ? g->L.total : 0; }
extern "C" int32_t eg_generator_num_weights(const EgGenerator* g) { return g ? (int32_t)g->L.entries.size() : 0; }
extern "C" int eg_generator_weight_entry(const EgGenerator* g, int32_t i, EgWeightEntry* out) {
    EG_REQUIRE(g && out && i >= 0 && i < (int)g->L.entries.size(), EG_ERR_BAD_ARG, "eg_generator_weight_entry: index %d", i);
    *out = g->L.entries[i];
    return EG_OK;
}
extern "C" int64_t eg_generator_workspace_bytes(const EgGenerator* g, int32_t batch) {
    if (!g || batch <= 0) return 0;
    return carve(g, batch).total;
}

extern "C" int eg_generator_forward(const EgGenerator* g, const float* arena, int32_t B, const float* spec, const int64_t* text,
                                    const float* prior, const float* sampled, float* pose, float* emotion_feature,
                                    float* semantic_feature, float* emotion_prediction, float* text_embedding, void* ws,
                                    int64_t ws_bytes, void* stream) {
    EG_REQUIRE(g && arena && spec && text && prior && ws, EG_ERR_BAD_ARG, "eg_generator_forward: null pointer");
    EG_REQUIRE(B > 0, EG_ERR_BAD_ARG, "eg_generator_forward: batch=%d", B);
    const GenWs w = carve(g, B);
    EG_REQUIRE(ws_bytes >= w.total, EG_ERR_WORKSPACE, "eg_generator_forward: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)w.total);
    EG_REQUIRE(eg_aligned16(ws) && eg_aligned16(arena) && eg_aligned16(spec), EG_ERR_ALIGN, "eg_generator_forward: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const EgGeneratorConfig& c = g->cfg;
    const int F = c.frames, D = c.d_model, rows = B * F, prec = c.precision;

    float* txt = text_embedding ? text_embedding : P(ws, w.t_out);
    hipStream_t s_text = st, s_prior = st;
    if (g->concurrent) {
        EG_TRY(g->ensure_streams());
        s_text = g->side[0]; s_prior = g->side[1];
        EG_HIP_TRY(hipEventRecord(g->ev_fork, st), "branch fork");
        EG_HIP_TRY(hipStreamWaitEvent(s_text, g->ev_fork, 0), "branch fork");
        EG_HIP_TRY(hipStreamWaitEvent(s_prior, g->ev_fork, 0), "branch fork");
    }
    EG_TRY(run_text(g, arena, text, txt, w, ws, B, s_text));
    EG_TRY(run_prior(g, arena, prior, w, ws, B, s_prior));
    if (g->concurrent) {
        EG_HIP_TRY(hipEventRecord(g->ev_join[0], s_text), "branch join");
        EG_HIP_TRY(hipEventRecord(g->ev_join[1], s_prior), "branch join");
    }
    EG_TRY(run_audio_tower(g, arena, spec, w, ws, B, st));
    const float* feat = P(ws, w.afeat);
    float* emo = emotion_feature ? emotion_feature : P(ws, w.emo);
    float* sem = semantic_feature ? semantic_feature : P(ws, w.sem);
    // emotion_proj.0 | semantic_proj.0 fused: [rows, 2D] = (emotion hidden | semantic hidden)
    const Act afeat = act(P(ws, w.afeat), D, P(ws, w.im_a[1]), D), es = act(P(ws, w.emo_t), 2 * D, P(ws, w.im_a[2]), 2 * D);
    if (g->fold) {
        EG_TRY(lin(g, arena, g->f_emo, afeat, 0, act(emo, D), true, rows, 0, nullptr, 0, st));
        EG_TRY(lin(g, arena, g->f_sem, afeat, 0, act(sem, D), true, rows, 0, nullptr, 0, st));
    } else {
        EG_TRY(lin(g, arena, g->emosem0, afeat, 0, es, false, rows, 0, nullptr, 0, st));
        EG_TRY(lin(g, arena, g->emo2, es, 0, act(emo, D), true, rows, 0, nullptr, 0, st));
        EG_TRY(lin(g, arena, g->sem2, es, D, act(sem, D), true, rows, 0, nullptr, 0, st));
    }
    // emotion classifier header on emotion_feature.reshape(B, F*D)  (:592)
    {
        const int K0 = F * D;
        EG_TRY(eg_linear_splitk(emo, K0, arena + g->cls[0].w, g->cls[0].kpad, arena + g->cls[0].b, P(ws, w.cls_h[0]), D, B, D, K0, 1,
                                F, P(ws, w.cls_part), prec, st));
        EG_TRY(run_linear(arena, g->cls[1], P(ws, w.cls_h[0]), D, P(ws, w.cls_h[1]), 256, B, 1, nullptr, 0, prec, st));
        EG_TRY(run_linear(arena, g->cls[2], P(ws, w.cls_h[1]), 256, P(ws, w.cls_h[2]), 64, B, 1, nullptr, 0, prec, st));
        float* pred = emotion_prediction ? emotion_prediction : P(ws, w.cls_out);
        EG_TRY(run_linear(arena, g->cls[3], P(ws, w.cls_h[2]), 64, pred, 8, B, 0, nullptr, 0, prec, st));
    }
    EG_TRY(egi_add(sampled ? sampled : emo, sem, P(ws, w.fus_in), (size_t)rows * D, D, 0, st));
    float* pose_out = pose ? pose : P(ws, w.pose);
    if (g->concurrent) {        // join: the decoder needs the prior encoding; the caller's stream must also cover the text branch
        EG_HIP_TRY(hipStreamWaitEvent(st, g->ev_join[1], 0), "branch join");
        EG_HIP_TRY(hipStreamWaitEvent(st, g->ev_join[0], 0), "branch join");
    }
    return run_transformer(g, arena, P(ws, w.fus_in), act(P(ws, w.prior_enc), D, P(ws, w.im_p[1]), D), pose_out, w, ws, B, st);
}

extern "C" int64_t eg_generator_draws_workspace_bytes(const EgGenerator* g, int32_t batch, int32_t draws) {
    if (!g || batch <= 0 || draws <= 0) return 0;
    return carve(g, batch, batch * draws).total;
}

// BASELINE config 5: audio/semantic tower once per clip, fusion->enc->dec->post for every sampled emotion map.
extern "C" int eg_generator_forward_draws(const EgGenerator* g, const float* arena, int32_t B, int32_t R, const float* spec,
                                          const float* prior, const float* sampled, float* pose, void* ws, int64_t ws_bytes,
                                          void* stream) {
    EG_REQUIRE(g && arena && spec && prior && sampled && pose && ws, EG_ERR_BAD_ARG, "eg_generator_forward_draws: null pointer");
    EG_REQUIRE(B > 0 && R > 0, EG_ERR_BAD_ARG, "eg_generator_forward_draws: batch=%d draws=%d", B, R);
    const GenWs w = carve(g, B, B * R);
    EG_REQUIRE(ws_bytes >= w.total, EG_ERR_WORKSPACE, "eg_generator_forward_draws: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)w.total);
    hipStream_t st = (hipStream_t)stream;
    const EgGeneratorConfig& c = g->cfg;
    const int F = c.frames, D = c.d_model, rows = B * F, prec = c.precision;
    EG_TRY(run_audio_tower(g, arena, spec, w, ws, B, st));
    EG_TRY(run_prior(g, arena, prior, w, ws, B, st));
    const Act afeat = act(P(ws, w.afeat), D, P(ws, w.im_a[1]), D), es = act(P(ws, w.emo_t), 2 * D, P(ws, w.im_a[2]), 2 * D);
    if (g->fold) {
        EG_TRY(lin(g, arena, g->f_sem, afeat, 0, act(P(ws, w.sem), D), true, rows, 0, nullptr, 0, st));
    } else {
        EG_TRY(lin(g, arena, g->emosem0, afeat, 0, es, false, rows, 0, nullptr, 0, st));
        EG_TRY(lin(g, arena, g->sem2, es, D, act(P(ws, w.sem), D), true, rows, 0, nullptr, 0, st));
    }
    // fusion_in[(b,r,f)] = sampled[(b,r,f)] + semantic[(b,f)];  decoder target stream = prior_enc[b] for every draw
    EG_TRY(egi_add_bcast(sampled, P(ws, w.sem), P(ws, w.fus_in), (size_t)B * R * F, D, F, R, st));
    EG_TRY(egi_add_bcast(nullptr, P(ws, w.prior_enc), P(ws, w.prior_rep), (size_t)B * R * F, D, F, R, st));
    return run_transformer(g, arena, P(ws, w.fus_in), act(P(ws, w.prior_rep), D), pose, w, ws, B * R, st);
}

extern "C" int eg_generator_tap(const EgGenerator* g, int32_t batch, void* ws, const char* name, float** d_ptr, int64_t* numel) {
    EG_REQUIRE(g && ws && name && d_ptr && numel && batch > 0, EG_ERR_BAD_ARG, "eg_generator_tap: null pointer");
    const GenWs w = carve(g, batch);
    const EgGeneratorConfig& c = g->cfg;
    const int64_t BFD = (int64_t)batch * c.frames * c.d_model;
    const std::string n(name);
    int64_t off = -1, cnt = 0;
    if (n == "audio_feat") { off = w.afeat; cnt = BFD; }
    else if (n == "prior_enc") { off = w.prior_enc; cnt = BFD; }
    else if (n == "fusion") { off = w.fusion; cnt = BFD; }
    else if (n == "enc_out") { off = w.xa; cnt = BFD; }
    else if (n == "audio_map") { off = w.amap; cnt = (int64_t)batch * c.frames * g->HW3; }
    else if (n == "mha_out") { off = w.xb; cnt = BFD; }                     // last MultiHeadAttention output (post LayerNorm)
    else if (n == "attn_q") { off = w.q; cnt = BFD; }                       // last cross-attention query projection
    else if (n == "attn_qkv") { off = w.qkv; cnt = 3 * BFD; }               // last Q|K|V (self) or K|V (cross) projection
    else if (n == "attn_out") { off = w.ao; cnt = BFD; }                    // last attention output (heads concatenated)
    else if (n == "proj") { off = w.proj; cnt = BFD; }                      // last pre-LayerNorm sum (fc / w_2 output + residual)
    else if (n == "dec_out") { off = w.fus_h; cnt = BFD; }
    else if (g->keep_taps && n == "stem") { off = w.tap_stem; cnt = (int64_t)batch * g->H1 * g->W1 * 32; }
    else if (g->keep_taps && n == "layer1") { off = w.tap_l[0]; cnt = (int64_t)batch * g->H1 * g->W1 * 32; }
    else if (g->keep_taps && n == "layer2") { off = w.tap_l[1]; cnt = (int64_t)batch * g->H2 * g->W2 * 64; }
    else if (g->keep_taps && n == "layer3") { off = w.tap_l[2]; cnt = (int64_t)batch * g->H3 * g->W3 * 128; }
    EG_REQUIRE(off >= 0, EG_ERR_BAD_ARG, "eg_generator_tap: unknown or disabled tap '%s'", name);
    *d_ptr = P(ws, off);
    *numel = cnt;
    return EG_OK;
}

// =============================================== CVAE ========================================================
struct EgCvae {
    EgCvaeConfig cfg;
    Layout L;
    struct Conv { int64_t w, b, s, t; int cin, cout, k, stride, pad; bool act; };
    Conv enc[4], dec_conv[3];
    struct ConvT { int64_t w, b, s, t; int cin, cout; } dec_t[2];
    struct Lin { int64_t w, b; int in, out; } py0, py2, mu0, mu2, var0, var2, fz0, fz2;
};

namespace {
EgCvae::Lin cvae_lin(Layout& L, const std::string& p, int out, int in) {
    EgCvae::Lin l;
    l.in = in; l.out = out;
    l.w = L.raw(p + ".weight", (int64_t)out * in);
    l.b = L.raw(p + ".bias", out);
    return l;
}
EgCvae::Conv cvae_conv(Layout& L, const std::string& seq, int idx, int cout, int cin, int k, int stride, int pad, bool act) {
    EgCvae::Conv c;
    c.cin = cin; c.cout = cout; c.k = k; c.stride = stride; c.pad = pad; c.act = act; c.s = c.t = -1;
    const std::string p = seq + "." + std::to_string(idx);
    c.w = L.raw(p + ".weight", (int64_t)cout * cin * k);
    c.b = L.raw(p + ".bias", cout);
    if (act) add_bn(L, seq + "." + std::to_string(idx + 2), cout, cout, c.s, c.t);
    return c;
}
struct CvaeWs { int64_t py_h, zy, fz_h, z0, d1, d2, d3, d4, e1, e2, e3, e4, mu_h, var_h, z, total; };
CvaeWs cvae_carve(const EgCvae* c, int n) {
    CvaeWs w;
    Carver cv;
    const int F = c->cfg.frames, D = c->cfg.d_model, Q = D / 4;
    w.py_h = cv.take((int64_t)n * 16); w.zy = cv.take((int64_t)n * 64); w.fz_h = cv.take((int64_t)n * 128); w.z0 = cv.take((int64_t)n * 4 * Q);
    w.d1 = cv.take((int64_t)n * 8 * Q * 2); w.d2 = cv.take((int64_t)n * 16 * D); w.d3 = cv.take((int64_t)n * 32 * D); w.d4 = cv.take((int64_t)n * F * D);
    w.e1 = cv.take((int64_t)n * 32 * D); w.e2 = cv.take((int64_t)n * 16 * D); w.e3 = cv.take((int64_t)n * 8 * D / 2); w.e4 = cv.take((int64_t)n * 4 * Q);
    w.mu_h = cv.take((int64_t)n * 128); w.var_h = cv.take((int64_t)n * 128); w.z = cv.take((int64_t)n * 32);
    w.total = cv.off;
    return w;
}
// fusion_z_posterior + Decoder (BEAT_CVAE.py:355-369,377-381,444-446); zy [n,64] already in workspace
int cvae_decode(const EgCvae* c, const float* A, const CvaeWs& w, void* ws, int n, float* out, hipStream_t st) {
    const int F = c->cfg.frames, D = c->cfg.d_model, Q = D / 4;
    EG_TRY(egi_small_linear(P(ws, w.zy), 64, A + c->fz0.w, A + c->fz0.b, P(ws, w.fz_h), 128, n, 64, 128, st));
    EG_TRY(egi_small_linear(P(ws, w.fz_h), 128, A + c->fz2.w, A + c->fz2.b, P(ws, w.z0), 4 * Q, n, 128, 4 * Q, st));
    EG_TRY(egi_convt1d(P(ws, w.z0), A + c->dec_t[0].w, A + c->dec_t[0].b, A + c->dec_t[0].s, A + c->dec_t[0].t, P(ws, w.d1), n, 4, 8, Q, st));
    EG_TRY(egi_convt1d(P(ws, w.d1), A + c->dec_t[1].w, A + c->dec_t[1].b, A + c->dec_t[1].s, A + c->dec_t[1].t, P(ws, w.d2), n, 8, 16, 2 * Q, st));
    const EgCvae::Conv* dc = c->dec_conv;
    EG_TRY(egi_conv1d(P(ws, w.d2), A + dc[0].w, A + dc[0].b, A + dc[0].s, A + dc[0].t, P(ws, w.d3), n, 16, 32, D, 3, 1, 1, 1, st));
    EG_TRY(egi_conv1d(P(ws, w.d3), A + dc[1].w, A + dc[1].b, A + dc[1].s, A + dc[1].t, P(ws, w.d4), n, 32, F, D, 3, 1, 1, 1, st));
    return egi_conv1d(P(ws, w.d4), A + dc[2].w, A + dc[2].b, nullptr, nullptr, out, n, F, F, D, 3, 1, 1, 0, st);
}
}  // namespace

extern "C" int eg_cvae_default_config(EgCvaeConfig* c) {
    EG_REQUIRE(c, EG_ERR_BAD_ARG, "null config");
    memset(c, 0, sizeof(*c));
    c->frames = 60; c->d_model = 512; c->latent = 32; c->n_classes = 8;
    return EG_OK;
}

extern "C" int eg_cvae_create(const EgCvaeConfig* cfg, EgCvae** out) {
    EG_REQUIRE(cfg && out, EG_ERR_BAD_ARG, "eg_cvae_create: null pointer");
    EG_REQUIRE(cfg->latent == 32 && cfg->n_classes == 8 && cfg->d_model % 16 == 0 && cfg->frames > 0 && cfg->frames <= 256, EG_ERR_UNSUPPORTED,
               "eg_cvae_create: unsupported config");
    EgCvae* c = new EgCvae();
    c->cfg = *cfg;
    Layout& L = c->L;
    const int F = cfg->frames, Q = cfg->d_model / 4;
    c->enc[0] = cvae_conv(L, "Encoder", 0, 32, F, 3, 1, 1, true);
    c->enc[1] = cvae_conv(L, "Encoder", 3, 16, 32, 3, 1, 1, true);
    c->enc[2] = cvae_conv(L, "Encoder", 6, 8, 16, 5, 2, 2, true);
    c->enc[3] = cvae_conv(L, "Encoder", 9, 4, 8, 5, 2, 2, true);
    c->py0 = cvae_lin(L, "Posterior_Y_embedding.0", 16, 8); c->py2 = cvae_lin(L, "Posterior_Y_embedding.2", 32, 16);
    c->mu0 = cvae_lin(L, "fc_mu.0", 128, 4 * Q); c->mu2 = cvae_lin(L, "fc_mu.2", 32, 128);
    c->var0 = cvae_lin(L, "fc_var.0", 128, 4 * Q); c->var2 = cvae_lin(L, "fc_var.2", 32, 128);
    for (int i = 0; i < 2; ++i) {
        EgCvae::ConvT& t = c->dec_t[i];
        t.cin = i ? 8 : 4; t.cout = i ? 16 : 8;
        const std::string p = "Decoder." + std::to_string(3 * i);
        t.w = L.raw(p + ".weight", (int64_t)t.cin * t.cout * 3);
        t.b = L.raw(p + ".bias", t.cout);
        add_bn(L, "Decoder." + std::to_string(3 * i + 2), t.cout, t.cout, t.s, t.t);
    }
    c->dec_conv[0] = cvae_conv(L, "Decoder", 6, 32, 16, 3, 1, 1, true);
    c->dec_conv[1] = cvae_conv(L, "Decoder", 9, F, 32, 3, 1, 1, true);
    c->dec_conv[2] = cvae_conv(L, "Decoder", 12, F, F, 3, 1, 1, false);
    c->fz0 = cvae_lin(L, "fusion_z_posterior.0", 128, 64); c->fz2 = cvae_lin(L, "fusion_z_posterior.2", 4 * Q, 128);
    L.total = eg_round_up(L.total, 16);
    *out = c;
    return EG_OK;
}
extern "C" void eg_cvae_destroy(EgCvae* c) { delete c; }
extern "C" int64_t eg_cvae_arena_floats(const EgCvae* c) { return c ? c->L.total : 0; }
extern "C" int32_t eg_cvae_num_weights(const EgCvae* c) { return c ? (int32_t)c->L.entries.size() : 0; }
extern "C" int eg_cvae_weight_entry(const EgCvae* c, int32_t i, EgWeightEntry* out) {
    EG_REQUIRE(c && out && i >= 0 && i < (int)c->L.entries.size(), EG_ERR_BAD_ARG, "eg_cvae_weight_entry: index %d", i);
    *out = c->L.entries[i];
    return EG_OK;
}
extern "C" int64_t eg_cvae_workspace_bytes(const EgCvae* c, int32_t n) { return (c && n > 0) ? cvae_carve(c, n).total : 0; }

extern "C" int eg_cvae_sample(const EgCvae* c, const float* A, int32_t n, const float* y, const float* z, float* out, void* ws,
                              int64_t ws_bytes, void* stream) {
    EG_REQUIRE(c && A && y && z && out && ws && n > 0, EG_ERR_BAD_ARG, "eg_cvae_sample: null pointer");
    const CvaeWs w = cvae_carve(c, n);
    EG_REQUIRE(ws_bytes >= w.total, EG_ERR_WORKSPACE, "eg_cvae_sample: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // post_y = Posterior_Y_embedding(y); zy = cat(z, post_y)  (BEAT_CVAE.py:440-443)
    EG_TRY(egi_small_linear(y, 8, A + c->py0.w, A + c->py0.b, P(ws, w.py_h), 16, n, 8, 16, st));
    EG_TRY(egi_small_linear(P(ws, w.py_h), 16, A + c->py2.w, A + c->py2.b, P(ws, w.zy) + 32, 64, n, 16, 32, st));
    EG_TRY(egi_copy2d(z, 32, P(ws, w.zy), 64, n, 32, st));
    return cvae_decode(c, A, w, ws, n, out, st);
}

extern "C" int eg_cvae_forward(const EgCvae* c, const float* A, int32_t n, const float* x, const float* y, const float* eps, float* recon,
                               float* mu, float* logvar, void* ws, int64_t ws_bytes, void* stream) {
    EG_REQUIRE(c && A && x && y && eps && recon && mu && logvar && ws && n > 0, EG_ERR_BAD_ARG, "eg_cvae_forward: null pointer");
    const CvaeWs w = cvae_carve(c, n);
    EG_REQUIRE(ws_bytes >= w.total, EG_ERR_WORKSPACE, "eg_cvae_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int F = c->cfg.frames, D = c->cfg.d_model, Q = D / 4;
    const EgCvae::Conv* e = c->enc;
    EG_TRY(egi_conv1d(x, A + e[0].w, A + e[0].b, A + e[0].s, A + e[0].t, P(ws, w.e1), n, F, 32, D, 3, 1, 1, 1, st));
    EG_TRY(egi_conv1d(P(ws, w.e1), A + e[1].w, A + e[1].b, A + e[1].s, A + e[1].t, P(ws, w.e2), n, 32, 16, D, 3, 1, 1, 1, st));
    EG_TRY(egi_conv1d(P(ws, w.e2), A + e[2].w, A + e[2].b, A + e[2].s, A + e[2].t, P(ws, w.e3), n, 16, 8, D, 5, 2, 2, 1, st));
    EG_TRY(egi_conv1d(P(ws, w.e3), A + e[3].w, A + e[3].b, A + e[3].s, A + e[3].t, P(ws, w.e4), n, 8, 4, D / 2, 5, 2, 2, 1, st));
    EG_TRY(egi_small_linear(P(ws, w.e4), 4 * Q, A + c->mu0.w, A + c->mu0.b, P(ws, w.mu_h), 128, n, 4 * Q, 128, st));
    EG_TRY(egi_small_linear(P(ws, w.mu_h), 128, A + c->mu2.w, A + c->mu2.b, mu, 32, n, 128, 32, st));
    EG_TRY(egi_small_linear(P(ws, w.e4), 4 * Q, A + c->var0.w, A + c->var0.b, P(ws, w.var_h), 128, n, 4 * Q, 128, st));
    EG_TRY(egi_small_linear(P(ws, w.var_h), 128, A + c->var2.w, A + c->var2.b, logvar, 32, n, 128, 32, st));
    EG_TRY(eg_reparameterize(mu, logvar, eps, P(ws, w.z), (int64_t)n * 32, st));
    EG_TRY(egi_small_linear(y, 8, A + c->py0.w, A + c->py0.b, P(ws, w.py_h), 16, n, 8, 16, st));
    EG_TRY(egi_small_linear(P(ws, w.py_h), 16, A + c->py2.w, A + c->py2.b, P(ws, w.zy) + 32, 64, n, 16, 32, st));
    EG_TRY(egi_copy2d(P(ws, w.z), 32, P(ws, w.zy), 64, n, 32, st));
    return cvae_decode(c, A, w, ws, n, recon, st);
}

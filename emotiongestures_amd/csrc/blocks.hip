// Block-level operators with individually passed weights: the reference's L2 blocks
// (Full_model/SubLayers.py:9-84, Full_model/tcn.py:16-64) as single C-ABI calls.  Used by the module-level
// host mirrors and the per-block parity tests; the whole-model path (generator.hip) launches the same kernels.
//
// Weight pointers are EG_PACK_LINEAR images ([Npad][Kpad] fp32, followed by the bf16 hi/lo images); with
// precision == EG_PREC_F32 only the fp32 image is read, so a raw nn.Linear weight [N,K] (K % 4 == 0) also works.
#include "common.h"

#define EG_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc) return _rc;      \
    } while (0)

static inline float* WP(void* ws, int64_t floats) { return reinterpret_cast<float*>(ws) + floats; }

extern "C" int64_t eg_mha_workspace_bytes(int32_t batch, int32_t lq, int32_t lk, int32_t d_model, int32_t heads) {
    const int64_t inner = (int64_t)heads * 64;
    return (int64_t)sizeof(float) * ((int64_t)batch * lq * (2 * inner + d_model) + (int64_t)batch * lk * inner * 2) + 1024;
}

extern "C" int eg_multi_head_attention(const float* xq, const float* xkv, const float* wq, const float* wk, const float* wv,
                                       const float* wo, const float* ln_g, const float* ln_b, float* out, float* attn,
                                       int32_t batch, int32_t lq, int32_t lk, int32_t d_model, int32_t heads,
                                       int32_t precision, void* workspace, int64_t workspace_bytes, void* stream) {
    EG_REQUIRE(xq && xkv && wq && wk && wv && wo && ln_g && ln_b && out && workspace, EG_ERR_BAD_ARG, "eg_multi_head_attention: null pointer");
    EG_REQUIRE(heads > 0 && d_model > 0 && d_model % 4 == 0, EG_ERR_UNSUPPORTED, "eg_multi_head_attention: d_model=%d heads=%d", d_model, heads);
    EG_REQUIRE(workspace_bytes >= eg_mha_workspace_bytes(batch, lq, lk, d_model, heads), EG_ERR_WORKSPACE, "eg_multi_head_attention: workspace too small");
    const int D = d_model, I = heads * 64, rq = batch * lq, rk = batch * lk;    // d_k = d_v = 64 (the attention kernel's head width)
    const int dpad = (int)eg_round_up(D, 64), ipad = (int)eg_round_up(I, 64);   // K padding of the packed weights
    float* q = WP(workspace, 0);
    float* ao = q + (int64_t)rq * I;
    float* pr = ao + (int64_t)rq * I;
    float* k = pr + (int64_t)rq * D;
    float* v = k + (int64_t)rk * I;
    EG_TRY(eg_linear(xq, D, wq, dpad, nullptr, nullptr, nullptr, 0, q, I, rq, I, D, 0, 0, 0, precision, stream));
    EG_TRY(eg_linear(xkv, D, wk, dpad, nullptr, nullptr, nullptr, 0, k, I, rk, I, D, 0, 0, 0, precision, stream));
    EG_TRY(eg_linear(xkv, D, wv, dpad, nullptr, nullptr, nullptr, 0, v, I, rk, I, D, 0, 0, 0, precision, stream));
    EG_TRY(eg_attention(q, I, k, I, v, I, ao, I, attn, batch, heads, lq, lk, 64, precision, stream));
    EG_TRY(eg_linear(ao, I, wo, ipad, nullptr, xq, nullptr, D, pr, D, rq, D, I, 0, 0, 0, precision, stream));
    return eg_layernorm(pr, ln_g, ln_b, out, rq, D, 1e-6f, stream);
}

extern "C" int64_t eg_ffn_workspace_bytes(int32_t rows, int32_t d_model, int32_t d_inner) {
    return (int64_t)sizeof(float) * ((int64_t)rows * d_inner + (int64_t)rows * d_model) + 512;
}

extern "C" int eg_positionwise_ffn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* ln_g, const float* ln_b, float* out, int32_t rows, int32_t d_model,
                                   int32_t d_inner, int32_t precision, void* workspace, int64_t workspace_bytes, void* stream) {
    EG_REQUIRE(x && w1 && b1 && w2 && b2 && ln_g && ln_b && out && workspace, EG_ERR_BAD_ARG, "eg_positionwise_ffn: null pointer");
    EG_REQUIRE(workspace_bytes >= eg_ffn_workspace_bytes(rows, d_model, d_inner), EG_ERR_WORKSPACE, "eg_positionwise_ffn: workspace too small");
    float* h = WP(workspace, 0);
    float* pr = h + (int64_t)rows * d_inner;
    EG_TRY(eg_linear(x, d_model, w1, d_model, b1, nullptr, nullptr, 0, h, d_inner, rows, d_inner, d_model, 1, 0, 0, precision, stream));
    EG_TRY(eg_linear(h, d_inner, w2, d_inner, b2, x, nullptr, d_model, pr, d_model, rows, d_model, d_inner, 0, 0, 0, precision, stream));
    return eg_layernorm(pr, ln_g, ln_b, out, rows, d_model, 1e-6f, stream);
}

// TemporalConvNet (Full_model/tcn.py:49-64).  x, y [B, L, Cpad] channels-last with row stride Cpad = round_up(C, 64).
// w: per level two convs, each { tap0 image, tap1 image, bias[Npad] }, image = EG_PACK_WN_TAP ([Npad][Cpad] x2 floats),
// Npad = round_up(C, 64).
extern "C" int eg_tcn_forward(const float* x, const float* w, float* y, int32_t batch, int32_t len, int32_t c,
                              int32_t levels, int32_t precision, void* workspace, int64_t workspace_bytes, void* stream) {
    EG_REQUIRE(x && w && y && workspace && batch > 0 && len > 0 && levels > 0, EG_ERR_BAD_ARG, "eg_tcn_forward: null pointer or empty shape");
    EG_REQUIRE(c % 4 == 0, EG_ERR_UNSUPPORTED, "eg_tcn_forward: C=%d must be a multiple of 4", c);
    const int cpad = (int)eg_round_up(c, 64), npad = (int)eg_round_up(c, 64), rows = batch * len;
    const int64_t buf = (int64_t)rows * cpad;
    EG_REQUIRE(workspace_bytes >= (int64_t)sizeof(float) * buf * 4, EG_ERR_WORKSPACE, "eg_tcn_forward: workspace too small");
    const int64_t img = (int64_t)npad * cpad * 2, conv = 2 * img + npad;
    float* h1 = WP(workspace, 0);
    float* acc = h1 + buf;
    float* pp[2] = {acc + buf, acc + 2 * buf};
    const float* cur = x;
    for (int i = 0; i < levels; ++i) {
        const int d = 1 << i;
        const float* c1 = w + (int64_t)(2 * i) * conv;
        const float* c2 = c1 + conv;
        float* dst = (i == levels - 1) ? y : pp[i & 1];
        // conv(x)[t] = b + W0 x[t-d] + W1 x[t] (pad d, chomp d: tcn.py:12,18-24); relu; twice; relu(out + x) (:43-47)
        EG_TRY(eg_linear(cur, cpad, c1, cpad, c1 + 2 * img, nullptr, nullptr, 0, h1, cpad, rows, c, c, 0, d, len, precision, stream));
        EG_TRY(eg_linear(cur, cpad, c1 + img, cpad, nullptr, h1, nullptr, cpad, h1, cpad, rows, c, c, 1, 0, 0, precision, stream));
        EG_TRY(eg_linear(h1, cpad, c2, cpad, c2 + 2 * img, nullptr, nullptr, 0, acc, cpad, rows, c, c, 0, d, len, precision, stream));
        EG_TRY(eg_linear(h1, cpad, c2 + img, cpad, nullptr, acc, cur, cpad, dst, cpad, rows, c, c, 1, 0, 0, precision, stream));
        cur = dst;
    }
    return EG_OK;
}

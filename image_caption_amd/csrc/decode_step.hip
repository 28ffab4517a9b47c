// One greedy decoding step, composed from the library's own entry points: a host that is not Python (or a
// Python host that wants one call per token instead of a dozen) drives KV-cached captioning through this.
// Host code only: every launch goes through the C ABI declared in capdec.h.
#include "common.h"

namespace {

constexpr long ALIGN = 256;
inline long up(long n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

struct StepBuffers {
    void *hh, *hs, *qs, *os, *h2s, *a1s;      // compute dtype
    void* hhp;                                // k-panel image of hh (bf16)
    float *xs, *xm, *xo[2];                   // f32 residual stream of the appended row
    void* argmax_ws;
    long argmax_ws_bytes;
    long total;
};

StepBuffers carve(char* base, int B, int d, int ff, int V, int dtype) {
    const long e = dtype == CAPDEC_BF16 ? 2 : 4;
    StepBuffers s;
    long off = 0;
    auto take = [&](long bytes) { char* p = base ? base + off : nullptr; off += up(bytes); return (void*)p; };
    s.hh = take((long)B * d * e);
    s.hs = take((long)B * d * e);
    s.qs = take((long)B * d * e);
    s.os = take((long)B * d * e);
    s.h2s = take((long)B * d * e);
    s.a1s = take((long)B * ff * e);
    s.xs = (float*)take((long)B * d * 4);
    s.xm = (float*)take((long)B * d * 4);
    s.xo[0] = (float*)take((long)B * d * 4);
    s.xo[1] = (float*)take((long)B * d * 4);
    s.argmax_ws_bytes = capdec_logits_ce_workspace(B, V);
    s.argmax_ws = take(s.argmax_ws_bytes);
    s.hhp = take((long)B * d * e);
    s.total = off;
    return s;
}

capdec_gemm_desc linear(const void* x, const void* w, void* y, int M, int N, int K, int in_dtype, int out_dtype, const float* bias) {
    capdec_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.B = w; g.C = y;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.in_dtype = in_dtype; g.out_dtype = out_dtype;
    g.bias = bias;
    g.col_scale = 1.f;
    return g;
}

}  // namespace

#define CAPDEC_TRY(call)            \
    do {                            \
        const int rc__ = (call);    \
        if (rc__ != CAPDEC_OK) return rc__; \
    } while (0)

extern "C" long capdec_decode_step_workspace(int B, int d_model, int dim_ff, int vocab, int dtype) {
    if (B <= 0 || d_model <= 0 || dim_ff <= 0 || vocab <= 0) return 0;
    return carve(nullptr, B, d_model, dim_ff, vocab, dtype).total;
}

extern "C" int capdec_decode_step(const capdec_decode_desc* p, void* stream) {
    CAPDEC_CHECK(p != nullptr, "decode_step: null descriptor");
    CAPDEC_CHECK(p->B > 0 && p->d_model > 0 && p->nhead > 0 && p->dim_ff > 0 && p->vocab > 0 && p->n_blocks > 0, "decode_step: bad geometry");
    CAPDEC_CHECK(p->d_model % p->nhead == 0, "decode_step: d_model not divisible by nhead");
    CAPDEC_CHECK(p->dtype == CAPDEC_BF16 || p->dtype == CAPDEC_F32, "decode_step: bad dtype");
    CAPDEC_CHECK(p->blocks && p->head_w && p->emb && p->x_last && p->ids && p->done && p->out, "decode_step: null operand");
    CAPDEC_CHECK(p->t >= 0 && p->t < p->max_len, "decode_step: token index %d outside [0, %d)", p->t, p->max_len);
    const bool advance = p->t + 1 < p->max_len;
    CAPDEC_CHECK(!advance || (p->s_new > 0 && p->s_new < p->smax), "decode_step: appended row %d outside the cache (smax %d)", p->s_new, p->smax);
    const int B = p->B, d = p->d_model, ff = p->dim_ff, H = p->nhead, hd = d / H;
    const long need = capdec_decode_step_workspace(B, d, ff, p->vocab, p->dtype);
    if (!p->workspace || p->workspace_bytes < need) {
        capdec_set_error("decode_step: workspace %ld bytes < %ld", p->workspace_bytes, need);
        return CAPDEC_ERR_WORKSPACE;
    }
    CAPDEC_CHECK((((uintptr_t)p->workspace) & (ALIGN - 1)) == 0, "decode_step: workspace must be 256-byte aligned");
    const StepBuffers s = carve((char*)p->workspace, B, d, ff, p->vocab, p->dtype);
    const int cdt = p->dtype;

    // 1. score the newest row, pick the token, book-keep <eos>/<pad>
    // with the vocabulary matrix's image given, the newest rows are cast and laid out as an image in one pass
    const bool h_img = p->head_w_panels && cdt == CAPDEC_BF16 && (d % 32) == 0;
    if (h_img) CAPDEC_TRY(capdec_pack_kpanels(p->x_last, CAPDEC_F32, B, d, d, s.hhp, B, s.hh, d, stream));
    else CAPDEC_TRY(capdec_cast(p->x_last, CAPDEC_F32, s.hh, cdt, (long)B * d, stream));
    const capdec_kpanels panels = {h_img ? s.hhp : nullptr, h_img ? (long)B : 0L, p->head_w_panels, p->head_w_panel_rows, 0};
    CAPDEC_TRY(capdec_logits_argmax(s.hh, p->head_w, p->head_b, cdt, B, p->vocab, d, d, p->ld_head_w ? p->ld_head_w : d,
                                    p->ids, p->margin, p->head_w_panels ? &panels : nullptr, s.argmax_ws, s.argmax_ws_bytes, stream));
    CAPDEC_TRY(capdec_greedy_update(p->ids, p->done, p->out, B, p->max_len, p->t, p->eos_id, p->pad_id, stream));
    if (!advance) return CAPDEC_OK;

    // 2. append the token's embedding as row s_new and run it through every block against the cache
    CAPDEC_TRY(capdec_embed_rows(p->emb, p->ids, B, d, p->pos ? p->pos + (long)p->s_new * d : nullptr, s.xs, stream));
    const float* xcur = s.xs;
    const float scale = 1.0f / sqrtf((float)hd);
    for (int i = 0; i < p->n_blocks; ++i) {
        const capdec_block_params& w = p->blocks[i];
        CAPDEC_CHECK(w.kv_cache && w.in_proj_w && w.out_proj_w && w.linear1_w && w.linear2_w && w.norm1_w && w.norm1_b && w.norm2_w && w.norm2_b,
                     "decode_step: block %d has a null weight", i);
        CAPDEC_TRY(capdec_ln_fwd(xcur, w.norm1_w, w.norm1_b, p->ln_eps, B, d, s.hs, cdt, nullptr, nullptr, stream));
        capdec_gemm_desc g = linear(s.hs, w.in_proj_w, s.qs, B, 3 * d, d, cdt, cdt, w.in_proj_b);
        g.ldc = d;                                   // only the q columns reach C; K/V columns go to the cache
        g.col_scale = scale; g.col_scale_n = d;
        g.kv_cache = w.kv_cache; g.kv_q = d; g.kv_d = d; g.kv_B = B; g.kv_H = H; g.kv_hd = hd; g.kv_smax = p->smax; g.kv_s0 = p->s_new;
        CAPDEC_TRY(capdec_gemm_bias_act(&g, stream));
        CAPDEC_TRY(capdec_attn_decode(s.qs, w.kv_cache, s.os, cdt, B, H, hd, p->smax, p->s_new + 1, stream));
        g = linear(s.os, w.out_proj_w, s.xm, B, d, d, cdt, CAPDEC_F32, w.out_proj_b);
        g.residual = xcur; g.ldr = d;
        CAPDEC_TRY(capdec_gemm_bias_act(&g, stream));
        CAPDEC_TRY(capdec_ln_fwd(s.xm, w.norm2_w, w.norm2_b, p->ln_eps, B, d, s.h2s, cdt, nullptr, nullptr, stream));
        g = linear(s.h2s, w.linear1_w, s.a1s, B, ff, d, cdt, cdt, w.linear1_b);
        g.act = p->activation;
        CAPDEC_TRY(capdec_gemm_bias_act(&g, stream));
        float* xout = i + 1 == p->n_blocks ? p->x_last : s.xo[i & 1];
        g = linear(s.a1s, w.linear2_w, xout, B, d, ff, cdt, CAPDEC_F32, w.linear2_b);
        g.residual = s.xm; g.ldr = d;
        CAPDEC_TRY(capdec_gemm_bias_act(&g, stream));
        xcur = xout;
    }
    return CAPDEC_OK;
}

/*
 * libcapdec -- C ABI of the MI355X captioning-decoder hot path.
 *
 * The reference (ocmoney/Image_caption) has no FFI: its hot path is Python calling
 * stock torch.nn modules.  Each entry point below names the reference statements
 * (file:line) whose arithmetic it replaces.  All pointers are caller-owned DEVICE
 * pointers; every call is asynchronous on `stream` (a hipStream_t passed as void*)
 * and returns 0 or a negative CAPDEC_ERR_* code (text via capdec_last_error()).
 * The library allocates no device memory: scratch is passed in by the caller.
 *
 * Row convention: activations are [S*B, width] with row = s*B + b (the
 * reference's seq-first [S,B,d] tensors, img_text_concat.py:123, flattened).
 */
#ifndef CAPDEC_H
#define CAPDEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPDEC_ABI_VERSION 3

enum { CAPDEC_F32 = 0, CAPDEC_BF16 = 1 };
enum { CAPDEC_ACT_NONE = 0, CAPDEC_ACT_RELU = 1, CAPDEC_ACT_GELU = 2 };
/* attention visibility: key j visible to query i when ...
 *   NONE          always (what decoder.py:31-38 yields for S <= 12288)
 *   REFERENCE     j < 12288 or j <= i           (decoder.py:33-37 literally)
 *   PREFIX_CAUSAL j < prefix_len or j <= i      (no reference counterpart) */
enum { CAPDEC_MASK_NONE = 0, CAPDEC_MASK_REFERENCE = 1, CAPDEC_MASK_PREFIX_CAUSAL = 2 };
enum { CAPDEC_GATE_NONE = 0, CAPDEC_GATE_RELU_OUT = 1, CAPDEC_GATE_GELU_PRE = 2 };

enum {
    CAPDEC_OK = 0,
    CAPDEC_ERR_INVALID = -1,  /* bad argument / unsupported shape */
    CAPDEC_ERR_LAUNCH = -2,   /* HIP launch error */
    CAPDEC_ERR_WORKSPACE = -3 /* workspace too small */
};

int capdec_abi_version(void);
const char* capdec_last_error(void);

/* ------------------------------------------------------------------ dropout keys
 * Dropout masks are a pure function of (key0, key1, element index); the backward
 * entry points regenerate them from the same keys.  An element is dropped when its
 * 16-bit hash field is below floor(p * 65536): p is honoured to 1.5e-5. */
typedef struct {
    float p;        /* 0 disables */
    uint32_t key0;
    uint32_t key1;
} capdec_dropout;

/* Write the keep-mask (1.0 / 0.0) the kernels use for `n` consecutive element
 * indices starting at 0.  Used by tests to hand the same mask to the oracle. */
int capdec_dropout_mask(capdec_dropout d, long n, float* out, void* stream);

/* ------------------------------------------------------------------ prefix join + LayerNorm
 * Replaces img_text_concat.py:115-123 (cat dim=1 + transpose(0,1)) fused with
 * decoder.py:53 (norm1).  img [B,P,d], sos [B,1,d], cap [B,L,d] of `in_dtype`;
 * optional pos [S,d] f32 added per row (no reference counterpart).
 * Writes x [S*B,d] f32 (the residual stream), y = LN(x) of `y_dtype`,
 * mean/rstd [S*B] f32.  Any of x / y / mean / rstd may be NULL (y NULL = join only).
 * Rows of x below x_row0 are NOT written (a caller that needs the residual only for the
 * scored rows saves that traffic; pass 0 to keep every row). */
int capdec_concat_ln_fwd(const void* img, const void* sos, const void* cap, int in_dtype,
                         const float* pos, const float* gamma, const float* beta, float eps,
                         int B, int P, int L, int d,
                         float* x, long x_row0, void* y, int y_dtype, float* mean, float* rstd, void* stream);

/* The same fused step for an already joined, seq-first batch taken as a strided view of the
 * processed dataset: decoder.py:163 (`split_data['decoder_inputs'][:, i:batch_end, :]`, element
 * (s, b, j) at tgt[s*seq_stride + b*batch_stride + j]) + decoder.py:53 (norm1), without first
 * making the slice contiguous.  Outputs as for capdec_concat_ln_fwd with rows r = s*B + b. */
int capdec_seqfirst_ln_fwd(const void* tgt, int in_dtype, long seq_stride, long batch_stride,
                           const float* pos, const float* gamma, const float* beta, float eps,
                           int S, int B, int d,
                           float* x, long x_row0, void* y, int y_dtype, float* mean, float* rstd, void* stream);

/* decoder.py:53,58 (nn.LayerNorm, eps 1e-5): y = LN(x) for x [M,d] f32. */
int capdec_ln_fwd(const float* x, const float* gamma, const float* beta, float eps, long M, int d,
                  void* y, int y_dtype, float* mean, float* rstd, void* stream);

/* Backward of the above (autograd of decoder.py:53,58).  dy [M,d] of dy_dtype.
 * dx_out = dx_in (may be NULL = 0) + dLN/dx; dx_out may alias dx_in; dx_out NULL = only the
 * affine gradients are computed.  dgamma/dbeta [d] f32 are OVERWRITTEN.  partial: scratch of
 * capdec_ln_bwd_workspace(M,d) bytes. */
long capdec_ln_bwd_workspace(long M, int d);
int capdec_ln_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                  const float* mean, const float* rstd, long M, int d,
                  const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
                  void* partial, void* stream);
/* capdec_ln_bwd and the capdec_dropout_bwd_cast of its dx_out in one pass over the rows (decoder.py:58 below the
 * residual dropout of decoder.py:55, and decoder.py:53 below that of the previous block's decoder.py:63):
 * dx_out as above (required); g[m,c] = dx_out[m,c] * keep / (1-p) in g_dtype, the dropout element index being
 * m*d + c; gbias[c] = column sums of g (f32, OVERWRITTEN).  partial: capdec_ln_bwd_dropout_workspace(M,d) bytes. */
long capdec_ln_bwd_dropout_workspace(long M, int d);
int capdec_ln_bwd_dropout(const void* dy, int dy_dtype, const float* x, const float* gamma,
                          const float* mean, const float* rstd, long M, int d,
                          const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
                          capdec_dropout drop, void* g, int g_dtype, float* gbias,
                          void* partial, void* stream);
/* LayerNorm-1's affine folded into the packed QKV projection that follows it (decoder.py:53-54:
 * self_attn(norm1(x)) = xhat (W*gamma)^T + (b + W beta), xhat = the normalised rows): Wf[j,:] = W[j,:]*gamma,
 * bf[j] = bias[j] + sum_c W[j,c] beta[c].  W [N,K] in w_dtype, Wf [N,K] in wf_dtype (an f32 W may fold straight into a
 * bf16 Wf: one rounding); leading dimensions ldw, ldwf; bias may be NULL. */
int capdec_ln_fold_fwd(const void* W, int w_dtype, long ldw, const float* bias, const float* gamma, const float* beta,
                       int N, int K, void* Wf, int wf_dtype, long ldwf, float* bf, void* stream);
/* Backward of that pair (the part of decoder.py:189 through norm1 and in_proj) when no gradient flows to the rows:
 * from G = dY^T xhat [N,K] f32 (the weight-gradient product taken with the normalised rows) and s = column sums of dY [N]:
 * G <- gamma*G + beta s^T (= d in_proj_weight, in place); dgamma[c] = sum_j W[j,c] G[j,c]; dbeta[c] = sum_j W[j,c] s[j]
 * (both OVERWRITTEN).  Replaces the product dY W over all rows and the LayerNorm backward pass over them.
 * partial: capdec_ln_fold_bwd_workspace(N,K) bytes, 16-byte aligned. */
long capdec_ln_fold_bwd_workspace(int N, int K);
int capdec_ln_fold_bwd(float* G, long ldg, const float* s, const void* W, int w_dtype, long ldw, const float* gamma, const float* beta,
                       int N, int K, float* dgamma, float* dbeta, void* partial, void* stream);
/* Affine gradients of the LayerNorm that capdec_concat_ln_fwd applied (decoder.py:53 on the
 * joined rows of img_text_concat.py:115-123), reading the rows again from img / sos / cap
 * (+ pos) instead of a kept copy of the joined tensor.  dy [(P+1+L)*B, d]; mean/rstd as written
 * by the forward; dgamma/dbeta OVERWRITTEN; partial as for capdec_ln_bwd with M = (P+1+L)*B. */
int capdec_concat_ln_bwd_affine(const void* dy, int dy_dtype, const void* img, const void* sos, const void* cap, int in_dtype,
                                const float* pos, const float* gamma, const float* mean, const float* rstd,
                                int B, int P, int L, int d, float* dgamma, float* dbeta, void* partial, void* stream);

/* The same for a seq-first strided batch (see capdec_seqfirst_ln_fwd); dy [S*B, d]. */
int capdec_seqfirst_ln_bwd_affine(const void* dy, int dy_dtype, const void* tgt, int in_dtype, long seq_stride, long batch_stride,
                                  const float* pos, const float* gamma, const float* mean, const float* rstd,
                                  int S, int B, int d, float* dgamma, float* dbeta, void* partial, void* stream);

/* ------------------------------------------------------------------ GEMM + fused epilogue
 * C[m,n] = epilogue( sum_k A(m,k) * B(n,k) ).  Replaces every nn.Linear of
 * decoder.py:54,59,62,66 and their autograd products (dgrad / wgrad):
 *   forward  y = x W^T      : A=x [M,K],  B=W [N,K]            (a_kmajor=0, b_kmajor=0)
 *   dgrad    dx = dy W      : A=dy [M,K], B=W stored [K,N]     (b_kmajor=1)
 *   wgrad    dW = dy^T x    : A=dy stored [K,M], B=x stored [K,N]  (both kmajor)
 * Epilogue order: + bias[n]; * col_scale for n < col_scale_n; aux_out = value
 * (pre-activation); activation; * gate; dropout; + residual; (+ C when accumulate);
 * colsum[n] += value (atomic); store.  in_dtype BF16 runs on MFMA with f32
 * accumulation, F32 is an exact-f32 validation path. */
typedef struct {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;          /* in elements */
    int a_kmajor, b_kmajor;
    int in_dtype, out_dtype;
    const float* bias;           /* [N] or NULL */
    float col_scale; int col_scale_n;
    int act;                     /* CAPDEC_ACT_* */
    capdec_dropout drop;         /* on the output, element index m*N+n */
    const float* residual; long ldr;
    int accumulate;              /* f32 out only */
    /* gate: multiplies by the derivative of activation+dropout of a forward GEMM:
     *  RELU_OUT: gate tensor = saved forward output; factor = (g != 0) / (1 - gate_drop.p)
     *  GELU_PRE: gate tensor = saved pre-activation z; factor = gelu'(z) * keep / (1 - p) */
    const void* gate; long ldg; int gate_mode; int gate_dtype; capdec_dropout gate_drop;
    void* aux_out; long ldaux;   /* out_dtype, NULL = off */
    float* colsum;               /* [N] f32 += column sums of the stored values (atomic adds; with a gate-only bf16
                                  * epilogue and a workspace of ceil(M/64)*N*4 bytes: a deterministic two-stage sum) */
    /* K/V-cache scatter (greedy decoding): columns >= kv_q (kv_q = 0 or kv_d; N = kv_q + 2*kv_d)
     * go to kv_cache[b][h][k|v][kv_s0 + s][j], row m = s*kv_B + b; columns < kv_q go to C. */
    void* kv_cache; int kv_q, kv_d, kv_B, kv_H, kv_hd, kv_smax, kv_s0;
    /* optional scratch: a plain bf16 product (no epilogue term) whose tile grid is too small to fill the
     * GPU is split along K into as many f32 slabs of M*N*4 bytes as pay off and fit (at most 32) and
     * summed in a fixed order */
    void* workspace; long workspace_bytes;
    /* optional [M] f32: C = diag(row_scale) (A B); plain products only (no epilogue term, no accumulate) */
    const float* row_scale;
    /* optional, plain bf16 products only: kvec [K] bf16 (NULL with kvec_out given: a vector of ones), kvec_out [M] f32
     * (OVERWRITTEN) = sum_k A(m,k) * kvec[k], one more column of the product formed on the way where the kernel allows
     * (the bias gradient of a Linear next to its weight gradient: with ones, the column sums of the reduction-major
     * operand).  With a workspace, 32*M*4 bytes at its end are used for this. */
    const void* kvec; float* kvec_out;
    /* 0: A is the matrix lda describes.  R > 0 (plain bf16 products, K % 32 == 0): A's stored matrix X -- [M][K], or [K][M]
     * when a_kmajor -- lies as a column-panel image [cols / 32][R][32] (R >= its row count; element (r, c) at
     * ((c / 32) * R + r) * 32 + c % 32; lda unused): what capdec_logits_ce_fwd_ref writes with capdec_kpanels.probs_rows.
     * A row-major A read in 32-deep steps then comes in whole 128-byte lines. */
    long a_panel_rows;
    /* optional k-panel image of a row-major B ([K / 32][b_panel_rows >= N][32] bf16, capdec_pack_kpanels; NULL: none).  B
     * itself must still be given: kernels that can read the image do (large row-major products on the persistent
     * kernel, 5 % faster), the others read B.  Results are bit-identical. */
    const void* b_panels; long b_panel_rows;
} capdec_gemm_desc;

int capdec_gemm_bias_act(const capdec_gemm_desc* desc, void* stream);

/* ------------------------------------------------------------------ attention
 * Replaces nn.MultiheadAttention's core (decoder.py:54; torch functional.py
 * baddbmm -> softmax -> dropout -> bmm) for all heads of all sequences.
 * q/k/v/o are addressed as base + b*bstride + h*hstride + s*sstride + j (elements),
 * so both the packed [S*B,3d] projection output and a [B,H,S,hd] cache work.
 * q must already carry the hd^-0.5 scale.  Queries q_start..q_start+nq-1 of each
 * sequence are computed against keys 0..S-1.  lse [B*H*S] f32 receives
 * log-sum-exp of the masked scores (needed by the backward). */
typedef struct {
    const void* q; const void* k; const void* v; void* o;
    long q_bs, q_hs, q_ss;  long k_bs, k_hs, k_ss;  long v_bs, v_hs, v_ss;  long o_bs, o_hs, o_ss;
    int dtype;
    int B, H, S, hd;
    int q_start, nq;
    int mask_mode, prefix_len;
    capdec_dropout drop;     /* on probabilities; index ((b*H+h)*S+q)*S4+key, S4 = S rounded up to a multiple of 4 */
    float* lse;
} capdec_attn_desc;

int capdec_attn_fwd(const capdec_attn_desc* desc, void* stream);

/* Backward: given do (same addressing as o) produce dq, dk, dv (addressing given
 * by dq_*, dk_*, dv_* strides; bf16: dk / dv strides multiples of 8 elements, 16-byte stores).  delta [B*H*S] f32 scratch (rowsum(do*o)).
 * Queries outside [q_start, q_start+nq) contribute nothing; dq rows outside are
 * not written. */
typedef struct {
    capdec_attn_desc fwd;    /* same q,k,v,o,lse as the forward call */
    const void* dout; long do_bs, do_hs, do_ss;
    void* dq; long dq_bs, dq_hs, dq_ss;
    void* dk; long dk_bs, dk_hs, dk_ss;
    void* dv; long dv_bs, dv_hs, dv_ss;
    float dq_scale;          /* dq is multiplied by this (hd^-0.5 when q was pre-scaled) */
    float* delta;
    /* optional (bf16 only), both f32 and 16-byte aligned: column sums of dq per 16-row tile -> colsum_q [q_rows, H*hd]
     * (column h*hd + c) and of dk | dv -> colsum_kv [kv_rows, 2*H*hd], rows from capdec_attn_bwd_colsum_rows.  Summed
     * over their rows (capdec_colsum) they are the bias gradient of the packed QKV projection (decoder.py:54's
     * in_proj_bias) without another pass over dq / dk / dv. */
    float* colsum_q; float* colsum_kv;
} capdec_attn_bwd_desc;

int capdec_attn_bwd(const capdec_attn_bwd_desc* desc, void* stream);
/* rows of the two column-sum buffers above for a call with these shapes: B * ceil(nq/16) and B * ceil(S/16) */
void capdec_attn_bwd_colsum_rows(int B, int S, int nq, int hd, long* q_rows, long* kv_rows);

/* One query row per sequence against a K/V cache [B][H][2][smax][hd]
 * (greedy decoding; no reference counterpart: the reference has no generation).
 * q [B, H*hd] (scaled), o [B, H*hd]; keys 0..len-1 are visible. */
int capdec_attn_decode(const void* q, const void* kv_cache, void* o, int dtype,
                       int B, int H, int hd, int smax, int len, void* stream);

/* ------------------------------------------------------------------ vocabulary projection + cross-entropy
 * Replaces decoder.py:66 (output_projection) on the scored rows only, the
 * slice/permute/reshape of decoder.py:172-179, nn.CrossEntropyLoss() (mean over
 * all rows, no ignore_index, decoder.py:140) and the argmax accuracy of
 * decoder.py:182-183, without materialising logits.
 * h [M,d] (M = L*B scored rows, row = l*B + b), W [V,d], bias [V] f32,
 * labels [B,L] int64.  Outputs: lse [M], argmax [M] int32, loss_sum[0] (sum of
 * per-row losses), correct[0] (int32).  workspace: capdec_logits_ce_workspace bytes.
 * probs (optional, [M, ld_probs] of `dtype`, ld_probs a multiple of 8 in [V, roundup128(V)]):
 * receives the softmax numerators exp(z - max of the column's 64-wide slot) (columns V..ld_probs-1
 * zero); together with the workspace's per-slot maxima they let capdec_logits_ce_bwd_from_probs
 * form d(logits) without a second product over the vocabulary.
 * Tie rule of argmax / correct (and of capdec_logits_argmax's ids): exactly equal logits resolve to the lowest column,
 * as torch.argmax does.  In float32, and wherever a margin is asked for, nothing else is a tie.  The bf16 products of
 * at least 2^24 outputs run a kernel that carries the column in the low 6 mantissa bits of each logit: logits within
 * 64 ulp (2^-17 relative) of the row's maximum are then not told apart by value, and the lowest column among them
 * wins.  capdec_logits_ce_fwd_ref compares z - z_label instead of z, so its resolution is 2^-17 of the distance to the
 * label's logit; a row whose maximum IS the label's column within rounding counts as correct. */
long capdec_logits_ce_workspace(long M, int V);

/* k-panel images of the two operands of the vocabulary product (decoder.py:66: h [M,d] and output_projection.weight
 * [V,d]), optional everywhere they appear (NULL pointer, or NULL members: the row-major operands are read instead).
 * An image holds the same bf16 values as the matrix, laid out [d / 32 panels][rows][32]: the 32 reduction elements of a
 * row that one step of the product consumes are 64 consecutive bytes, and consecutive rows follow each other, so the
 * product's loads fetch whole 128-byte lines (row-major rows of 2*d bytes give it half lines).  Results are bit-identical
 * either way; made by capdec_pack_kpanels.  a / b: image of h / of W; a_rows >= M, b_rows >= V: rows per panel. */
typedef struct {
    const void* a; long a_rows;
    const void* b; long b_rows;
    long probs_rows;   /* capdec_logits_ce_fwd_ref only, 0 elsewhere.  0: probs is the matrix [M, ld_probs]; R >= M: probs is written
                        * as a column-panel image [ld_probs / 32][R][32] (ld_probs % 32 == 0): element (m, v) at
                        * ((v / 32) * R + m) * 32 + v % 32.  The data gradient then reads its 32-deep steps of d(logits) as whole
                        * lines too (capdec_gemm_desc.a_panel_rows); capdec_logits_ce_bwd_ref_prepare takes the same R. */
} capdec_kpanels;
/* panels[k / 32][r][k % 32] = bf16(src[r, k]) for r < rows, k < K (K % 32 == 0; src f32 or bf16 [rows, ld_src], rows
 * 16-byte aligned; panels 16-byte aligned, panel_rows >= rows rows per panel -- rows rows..panel_rows-1 are not written).
 * rowmajor (optional, bf16 [rows, ld_rm]) receives the plain cast of src in the same pass (capdec_cast). */
int capdec_pack_kpanels(const void* src, int src_dtype, long rows, int K, long ld_src,
                        void* panels, long panel_rows, void* rowmajor, long ld_rm, void* stream);
/* The same for `batch` matrices of one shape in one launch (the same weight of every block of a stack, which lie a
 * constant number of elements apart in a flat parameter buffer): matrix z starts src_batch_stride elements after matrix
 * z - 1, its image panels_batch_stride elements after the previous image.  No row-major copy. */
int capdec_pack_kpanels_batched(const void* src, int src_dtype, long rows, int K, long ld_src, long src_batch_stride,
                                void* panels, long panel_rows, long panels_batch_stride, int batch, void* stream);

int capdec_logits_ce_fwd(const void* h, const void* W, const float* bias, int dtype,
                         long M, int V, int d, long ldh, long ldw,
                         const int64_t* labels, int B, int L,
                         float* lse, int32_t* argmax, float* loss_sum, int32_t* correct,
                         void* probs, long ld_probs, const capdec_kpanels* panels,
                         void* workspace, long workspace_bytes, void* stream);

/* Backward of the mean loss w.r.t. logits (autograd of decoder.py:140,180,189) from what the
 * forward kept: probs_dl [M, ld] holds the numerators on entry and
 *   dl[m,n] = (softmax(z)[m,n] - [n == label[m]]) * gscale
 * on return (in place, same dtype).  fwd_workspace is the forward's workspace, untouched since;
 * dbias [V] (optional) is OVERWRITTEN with the column sums, through `workspace` of
 * capdec_logits_ce_bwd_workspace(M, V) bytes. */
int capdec_logits_ce_bwd_from_probs(void* probs_dl, int dtype, long M, int V, long ld,
                                    const int64_t* labels, int B, int L, const float* lse, float gscale,
                                    const void* fwd_workspace, long fwd_workspace_bytes,
                                    float* dbias, void* workspace, long workspace_bytes, void* stream);

/* The same gradient by RECOMPUTING the logits (for batches whose d(logits) does not fit in one
 * buffer), one vocabulary slab [v0, v0+nv):
 * dl[m, n-v0] = (exp(z[m,n] - lse[m]) - [n == label[m]]) * gscale, written to
 * dl [M, ld_dl] of `dtype` (columns nv..ld_dl-1 zero-filled); dbias[v0 .. v0+nv) is
 * OVERWRITTEN with the column sums (fixed-order reduction through `workspace` of
 * capdec_logits_ce_bwd_workspace(M, nv) bytes; both may be NULL).  The caller then forms
 * dH += dl W_slab and dW_slab = dl^T H with capdec_gemm_bias_act. */
long capdec_logits_ce_bwd_workspace(long M, int nv);
int capdec_logits_ce_bwd_slab(const void* h, const void* W, const float* bias, int dtype,
                              long M, int V, int d, long ldh, long ldw,
                              const int64_t* labels, int B, int L, const float* lse, float gscale,
                              int v0, int nv, void* dl, long ld_dl, float* dbias,
                              void* workspace, long workspace_bytes, void* stream);

/* capdec_logits_ce_fwd (decoder.py:66 on the scored rows + decoder.py:172-183) with the softmax numerators kept
 * against ONE reference value per row instead of one per 64-column slot: rowref[m] (out, [M] f32, 16-byte aligned) =
 * the label's logit, probs[m,v] = exp(z[m,v] - rowref[m]) (bf16, required; columns V..ld_probs-1 zero).  d(logits) is
 * then a row factor times the numerators (capdec_logits_ce_bwd_ref_prepare), so the backward products read probs as
 * it is and no pass turns it into d(logits).  A row whose maximum exceeds its label's logit by more than 50 (label
 * probability below e^-50) is redone against its maximum, rowref[m] holding that instead.  bf16 only, d % 32 == 0,
 * d >= 256.  lse / argmax / loss_sum / correct / workspace as for capdec_logits_ce_fwd. */
int capdec_logits_ce_fwd_ref(const void* h, const void* W, const float* bias, int dtype,
                             long M, int V, int d, long ldh, long ldw,
                             const int64_t* labels, int B, int L,
                             float* lse, int32_t* argmax, float* loss_sum, int32_t* correct,
                             void* probs, long ld_probs, float* rowref, const capdec_kpanels* panels,
                             void* workspace, long workspace_bytes, void* stream);
/* Backward of decoder.py:172 from those numerators: with g[m] = gscale * exp(rowref[m] - lse[m]),
 * d(logits)[m,v] = g[m] * probs'[m,v], probs' = probs except probs'[m,label] = probs[m,label] - exp(lse[m] - rowref[m])
 * (patched IN PLACE here).  Writes rowscale[m] = g[m] ([M] f32) and h_scaled[m,:] = g[m] * h[m,:] (bf16 [M, ld_hs]):
 *   d head.weight = probs'^T h_scaled,  d h = diag(rowscale) (probs' W)  (capdec_gemm_desc.row_scale),
 *   d head.bias = capdec_colsum_weighted(probs', rowscale), or, with rowscale_bf16 (optional [M] bf16 copy of g) as
 *   capdec_gemm_desc.kvec of the weight-gradient product, one more column of that product.
 * probs_panel_rows: 0 when probs is the matrix [M, ld_probs]; the R given to capdec_logits_ce_fwd_ref as
 * capdec_kpanels.probs_rows when it is the column-panel image [ld_probs / 32][R][32] (capdec_colsum_weighted reads the
 * matrix form only: with an image the bias gradient is the products' kvec column). */
int capdec_logits_ce_bwd_ref_prepare(void* probs, long ld_probs, long M, int V,
                                     const void* h, long ldh, int d,
                                     const int64_t* labels, int B, int L,
                                     const float* lse, const float* rowref, float gscale,
                                     void* h_scaled, long ld_hs, float* rowscale, void* rowscale_bf16,
                                     long probs_panel_rows, void* stream);
/* Greedy step: argmax_n(h W^T + bias) per row without writing logits.
 * out_ids [M] int64; optional margin [M] f32 = top1 - top2. workspace as above.  Ties: see capdec_logits_ce_fwd; with
 * margin != NULL the ids are exact argmaxes, without it two logits within 2^-17 relative of each other may resolve to
 * the lower column although the higher one is larger by less than that. */
int capdec_logits_argmax(const void* h, const void* W, const float* bias, int dtype,
                         long M, int V, int d, long ldh, long ldw,
                         int64_t* out_ids, float* margin, const capdec_kpanels* panels,
                         void* workspace, long workspace_bytes, void* stream);

/* ------------------------------------------------------------------ element-wise helpers */
/* dst = cast(src) for n elements */
int capdec_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
/* dst[c, r] = src[r, c] for a bf16 matrix [rows, cols] (leading dimensions in elements).  The data gradients of
 * decoder.py:189 read a weight [out, in] along `out`; on a transposed copy made once per step the second operand of
 * those products is row-major, which the product kernels read at twice the LDS rate of the transposing read. */
int capdec_transpose_bf16(const void* src, long ld_src, long rows, long cols, void* dst, long ld_dst, void* stream);
/* Column sums are computed in two deterministic stages (row stripes, then a fixed-order fold) and
 * need a scratch buffer of capdec_colsum_workspace(M,N) bytes. */
long capdec_colsum_workspace(long M, int N);
/* Backward of `x + dropout(y)` w.r.t. y (decoder.py:55,63): g[m,n] = dx[m,n] * keep / (1-p)
 * cast to g_dtype; colsum[n] (optional, f32, OVERWRITTEN) receives the column sums of g
 * (= bias gradient of the Linear that produced y); workspace only needed with colsum. */
int capdec_dropout_bwd_cast(const float* dx, long M, int N, capdec_dropout d,
                            void* g, int g_dtype, float* colsum,
                            void* workspace, long workspace_bytes, void* stream);
/* colsum[n] = sum_m x[m,n]  (bias gradients; OVERWRITTEN) */
int capdec_colsum(const void* x, int dtype, long M, int N, long ldx, float* colsum,
                  void* workspace, long workspace_bytes, void* stream);
/* colsum[n] = sum_m row_weight[m] * x[m,n]  (OVERWRITTEN; the vocabulary bias gradient of
 * capdec_logits_ce_fwd_ref's numerators); workspace as for capdec_colsum. */
int capdec_colsum_weighted(const void* x, int dtype, long M, int N, long ldx, const float* row_weight, float* colsum,
                           void* workspace, long workspace_bytes, void* stream);
/* capdec_colsum without a caller workspace.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The
 * stripe partials go to a fixed CAPDEC_COLSUM_STATIC_BYTES array inside the library's code object (one copy per GPU,
 * held until the process ends); the call allocates nothing and returns CAPDEC_ERR_WORKSPACE when
 * capdec_colsum_workspace(M, N) exceeds that size.  Results are bit-identical to capdec_colsum.  All calls on one GPU
 * share the array: calls that overlap in time on the same GPU (different streams) race.
 * The device that owns `stream` must be the current device (checked: CAPDEC_ERR_INVALID otherwise) -- the array's
 * address is per device.  The FIRST call on a device resolves that address (hipGetSymbolAddress, which may load the
 * code object): make it outside any stream capture; later calls only launch kernels and can be captured into a graph. */
#define CAPDEC_COLSUM_STATIC_BYTES (4L << 20)
int capdec_colsum_static(const void* x, int dtype, long M, int N, long ldx, float* colsum, void* stream);
/* rows of a token-embedding table: out[i,:] = table[ids[i],:] (+ pos row if given); out f32 */
int capdec_embed_rows(const float* table, const int64_t* ids, long n, int d, const float* pos_row,
                      float* out, void* stream);

/* A row-major matrix [rows, K] inside the flat buffer whose k-panel image (capdec_kpanels.b: the vocabulary matrix) the
 * update keeps current: it occupies elements first .. first + rows*K - 1 counted from `w` (first may be negative or the
 * matrix may extend past n when the call covers a slice of the buffer: only the elements inside the call are written). */
typedef struct {
    void* panels;            /* [K / 32][panel_rows][32] bf16, or NULL */
    long first;              /* a multiple of 4 */
    long rows; int K;        /* K a multiple of 32 */
    long panel_rows;         /* >= rows */
} capdec_adam_panels;
/* torch.optim.Adam(lr, betas, eps, weight_decay) with coupled L2 (decoder.py:135-139,190)
 * over a flat f32 parameter buffer; optionally refreshes a bf16 shadow copy used by
 * the MFMA kernels and (panels, may be NULL) the k-panel image of one matrix in it.
 * grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int capdec_adam_step(float* w, const float* g, float* m, float* v, long n, int step,
                     float lr, float beta1, float beta2, float eps, float weight_decay,
                     float grad_scale, void* w_bf16, const capdec_adam_panels* panels, void* stream);

/* Greedy bookkeeping: ids[b] <- pad where done[b]; done[b] |= ids[b]==eos; out[b*L+t] = ids[b]. */
int capdec_greedy_update(int64_t* ids, uint8_t* done, int64_t* out, int B, int L, int t,
                         int64_t eos_id, int64_t pad_id, void* stream);

/* ------------------------------------------------------------------ composed driver: one greedy step
 * No reference counterpart (the reference has no generation); the block arithmetic is that of
 * decoder.py:53-63 on one appended row, the scoring that of decoder.py:66 + an argmax.
 * One call = token t of every sequence:
 *   ids = argmax(x_last W_head^T + b_head); <eos>/<pad> book-keeping into out[:, t]  (as capdec_greedy_update);
 *   then, unless t is the last token: row s_new = emb[ids] (+ pos[s_new]) goes through every block against
 *   the K/V caches (its own K/V are appended at s_new) and x_last receives its hidden state.
 * `blocks` is a HOST array of n_blocks entries of device pointers: matrices in `dtype`, vectors f32,
 * kv_cache [B][H][2][smax][hd] in `dtype` already holding rows 0..s_new-1.
 * workspace: capdec_decode_step_workspace(...) bytes, 256-byte aligned; nothing is allocated. */
typedef struct {
    const float* norm1_w; const float* norm1_b;
    const void* in_proj_w; const float* in_proj_b;      /* [3d, d], [3d] */
    const void* out_proj_w; const float* out_proj_b;    /* [d, d], [d] */
    const float* norm2_w; const float* norm2_b;
    const void* linear1_w; const float* linear1_b;      /* [ff, d], [ff] */
    const void* linear2_w; const float* linear2_b;      /* [d, ff], [d] */
    void* kv_cache;
} capdec_block_params;

typedef struct {
    int B, d_model, nhead, dim_ff, vocab, n_blocks, smax;
    int dtype;                   /* CAPDEC_BF16 / CAPDEC_F32: matrices, caches and internal activations */
    int activation;              /* CAPDEC_ACT_* of the MLP */
    float ln_eps;
    const capdec_block_params* blocks;
    const void* head_w; const float* head_b; long ld_head_w;   /* [vocab, d_model]; ld 0 = d_model */
    const float* emb;            /* token-embedding table [vocab, d_model] f32 */
    const float* pos;            /* optional [smax, d_model] f32 positional rows, or NULL */
    float* x_last;               /* [B, d_model] f32: hidden state of the newest row (in / out) */
    int64_t* ids;                /* [B] out: the chosen tokens */
    uint8_t* done;               /* [B] in / out */
    int64_t* out;                /* [B, max_len] */
    float* margin;               /* optional [B]: top1 - top2 logit */
    int max_len, t, s_new;
    int64_t eos_id, pad_id;
    void* workspace; long workspace_bytes;
    const void* head_w_panels; long head_w_panel_rows;   /* optional k-panel image of head_w (capdec_kpanels.b), or NULL */
} capdec_decode_desc;

long capdec_decode_step_workspace(int B, int d_model, int dim_ff, int vocab, int dtype);
int capdec_decode_step(const capdec_decode_desc* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPDEC_H */

// C-ABI of libtitok_hip.so (include/titok_hip.h): argument checks, workspace carving and the launch
// sequences of the encoder / decoder towers.  Everything here only enqueues on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

static thread_local char g_err[512] = "";

void ttv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- measurement hook ----
int g_ttv_prof_class = 0;
thread_local int g_ttv_debug = 0;     // per host thread: a thread that forces a kernel variant (tests, A/B tools) does not change what another thread launches
long long* g_ttv_stamps = nullptr;   // diagnostics: device buffer for in-kernel clock stamps (ttv_debug_stamps)
static hipEvent_t* g_prof_start = nullptr;
static hipEvent_t* g_prof_stop = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;

TtvProfScope::TtvProfScope(int cls, hipStream_t stream) : slot(-1), s(stream) {
  if (g_ttv_prof_class != 0 && cls == g_ttv_prof_class && g_prof_n < g_prof_cap) {
    slot = g_prof_n++;
    (void)hipEventRecord(g_prof_start[slot], s);
  }
}
TtvProfScope::~TtvProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof_stop[slot], s);
}

#define TTV_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != TTV_OK) return rc__; \
  } while (0)

static inline int esize(int dtype) { return dtype == TTV_BF16 ? 2 : 4; }
static inline int64_t align_up(int64_t v) { return (v + 255) & ~(int64_t)255; }

struct TowerWs {
  char *x, *xn, *qkv, *ao, *h, *pa, *pb;
  char *xl, *aol;    // [sum_tokens, width]: the latent rows of x and of the attention output, compact (encoder, last layer)
  char *f8, *f8mx;   // block-scaled fp8 image of the running linear's input [L, max(width, inner)] and its E8M0 scales
  float* y32;
  float* rstd;     // [L] row statistic of a folded pre-norm (generic-width bf16 towers)
  int64_t total;
};

static TowerWs carve(const ttv_tower_dims* d, const ttv_batch* b, char* base) {
  const int64_t e = esize(d->dtype);
  const int64_t L = b->total_rows, P = b->sum_patches;
  const int64_t g = (int64_t)d->kv_heads * d->head_dim;
  const int64_t pd = (int64_t)d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  int64_t off = 0;
  TowerWs w;
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
  w.x = take(L * d->width * e);
  w.xn = take(L * d->width * e);
  w.qkv = take(L * (2 * d->width + 2 * g) * e);
  w.ao = take(L * d->width * e);
  w.y32 = (float*)take(L * d->width * 4);
  w.h = take(L * d->inner * e);
  w.pa = take(P * pd * e);        // encoder: gathered patches; decoder: proj_out output
  w.pb = take(P * d->width * e);  // encoder: proj_in output;   decoder: ln_post output
  w.rstd = (float*)take(L * 4);
  w.xl = take((int64_t)b->sum_tokens * d->width * e);
  w.aol = take((int64_t)b->sum_tokens * d->width * e);
  w.f8 = w.f8mx = nullptr;
  if (d->dtype == TTV_BF16 && d->width != 256 && d->width % 128 == 0) {      // the towers that can run the block-scaled fp8 linears
    const int64_t wide = d->width > d->inner ? d->width : d->inner;
    w.f8 = take(L * wide);
    w.f8mx = take(L * (wide / 128 + 4) * 4);
  }
  w.total = off;
  return w;
}

static int check_dims(const ttv_tower_dims* d, const ttv_batch* b) {
  TTV_CHECK_ARG(d && b, "null dims/batch");
  TTV_CHECK_ARG(d->dtype == TTV_BF16 || d->dtype == TTV_F32, "bad dtype %d", d->dtype);
  TTV_CHECK_ARG(d->head_dim == 64 && d->width == d->q_heads * 64, "width must be q_heads*64");
  TTV_CHECK_ARG(d->width % 64 == 0 && d->width <= 1024, "width %d unsupported", d->width);
  TTV_CHECK_ARG(d->inner % 32 == 0, "GEGLU inner dim must be a multiple of 32");
  TTV_CHECK_ARG(d->token_size >= 1 && d->token_size <= TTV_MAX_TOKEN, "token_size out of range");
  TTV_CHECK_ARG(((int64_t)d->pix_channels * d->patch_t * d->patch_h * d->patch_w) % 8 == 0, "patch vector length must be a multiple of 8");
  TTV_CHECK_ARG(b->n_clips > 0 && b->total_rows == b->sum_tokens + b->sum_patches, "inconsistent batch");
  return TTV_OK;
}

// One layer with all four linears on the block-scaled fp8 MFMA (BASELINE config #5; ttv_layer_weights.to_qkv_mx ...): every linear's
// input is quantised by k_quant_mx_fp8, the pre-norm gains live in the weight images and the rstd of the pre-norm is the activation's
// per-row factor in the GEMM epilogue; attention, the KEEL sums and norms stay bf16 / fp32 as in the bf16 tower.
static int run_layer_mx(const ttv_tower_dims* d, const ttv_layer_weights& lw, const ttv_batch* b, const TowerWs& ws, int i, bool& rstd_valid,
                        bool& xq_valid, bool attn_pipe, hipStream_t s) {
  const int L = b->total_rows, dm = d->width, g = d->kv_heads * d->head_dim, dt = d->dtype, nq = 2 * dm + 2 * g, I = d->inner;
  // Quantisation is fused into the producers where the producer owns whole 32-element blocks (TTV_MX_FUSED_QUANT=0: every operand by a
  // pass of its own, A/B): the KEEL post-norm kernel writes x also as its block-scaled image (ws.f8 / ws.f8mx), the w12 GEMM's GEGLU
  // epilogue writes h only as its image (into the bf16 h buffer and the xn buffer, both unused in this mode).  Left as passes: the
  // attention output (its kernel is untouched) and layer 0's two inputs (no norm kernel in front of them).
  static const bool fused_env = !(getenv("TTV_MX_FUSED_QUANT") && getenv("TTV_MX_FUSED_QUANT")[0] == '0');
  const bool fused_q = fused_env && !(g_ttv_debug & 2048);      // ttv_debug_set bit 11: the unfused sequence (tests: both must agree bit for bit)
  if (!rstd_valid) TTV_TRY(ttvk_row_rstd(ws.x, dt, dm, ws.rstd, L, dm, d->eps, s));
  if (!xq_valid) TTV_TRY(ttvk_quant_mx_fp8(ws.x, dt, dm, ws.f8, dm, ws.f8mx, nullptr, L, dm, s));
  xq_valid = false;
  GemmArgs a = {};
  a.dtype = dt; a.x = ws.f8; a.ldx = dm; a.w = lw.to_qkv_f8; a.ldw = dm; a.M = L; a.N = nq; a.K = dm; a.y = ws.qkv; a.ldy = nq;
  a.rope_cs = b->rope_cs; a.rope_q_end = dm; a.rope_k_begin = 2 * dm; a.rope_k_end = 2 * dm + g;
  TTV_TRY(ttvk_gemm_fp8(EPI_QKV_ROPE, a, ws.rstd, lw.to_qkv_f8_scale, s, ws.f8mx, lw.to_qkv_mx));
  const bool q_scaled = lw.qkv_q_prescaled != 0;
  if (fused_q && q_scaled && !b->qblocks_paired && !attn_pipe && d->head_dim == 64) {
    // the attention epilogue writes out_proj's operand itself (no bf16 output, no pass over it); ws.f8 is free: the qkv GEMM has read it
    TTV_TRY(ttvk_attention_mxout(ws.qkv, nq, ws.f8, ws.f8mx, (int)ttvk_mx_scale_ld(dm), b->cu_seqlens, b->qblocks, b->n_qblocks, d->q_heads,
                                 d->kv_heads, s));
  } else {
    // (no TTV_ATTN_ALLFULL unless the pipelined kernel is asked for: this is the A/B twin of the attention kernel with the fused MX
    // epilogue above, which is k_attn_bf16 - the two must produce the same bits, tested)
    TTV_TRY(ttvk_attention(ws.qkv, nq, ws.ao, dm, b->cu_seqlens, b->qblocks, b->n_qblocks, d->q_heads, d->kv_heads, d->head_dim,
                           TTV_ATTN_GATE | (b->qblocks_paired ? TTV_ATTN_PAIRED : 0) | (q_scaled ? TTV_ATTN_QSCALED : 0) |
                               ((b->qblocks_all_full && attn_pipe) ? TTV_ATTN_ALLFULL : 0) | (attn_pipe ? TTV_ATTN_PIPE : 0), dt, s));
    TTV_TRY(ttvk_quant_mx_fp8(ws.ao, dt, dm, ws.f8, dm, ws.f8mx, nullptr, L, dm, s));
  }
  GemmArgs o = {};
  o.dtype = dt; o.x = ws.f8; o.ldx = dm; o.w = lw.out_proj_f8; o.ldw = dm; o.M = L; o.N = dm; o.K = dm; o.resid = ws.x; o.ldr = dm;
  o.alpha = i == 0 ? 1.f : d->alpha; o.y = ws.x; o.ldy = dm;
  TTV_TRY(ttvk_gemm_fp8(EPI_RESID_T, o, nullptr, lw.out_proj_f8_scale, s, ws.f8mx, lw.out_proj_mx));
  if (i > 0) {
    TTV_TRY(ttvk_rmsnorm(ws.x, dt, dm, nullptr, ws.x, dt, dm, nullptr, lw.attn_post_ln, L, dm, d->eps, s, ws.rstd, fused_q ? ws.f8 : nullptr,
                         fused_q ? ws.f8mx : nullptr));
    if (!fused_q) TTV_TRY(ttvk_quant_mx_fp8(ws.x, dt, dm, ws.f8, dm, ws.f8mx, nullptr, L, dm, s));
  } else {
    TTV_TRY(ttvk_row_rstd(ws.x, dt, dm, ws.rstd, L, dm, d->eps, s));
    TTV_TRY(ttvk_quant_mx_fp8(ws.x, dt, dm, ws.f8, dm, ws.f8mx, nullptr, L, dm, s));
  }
  GemmArgs f = {};
  f.dtype = dt; f.x = ws.f8; f.ldx = dm; f.w = lw.w12_f8; f.ldw = dm; f.M = L; f.N = I; f.K = dm; f.y = ws.h; f.ldy = I;
  char* const hq = ws.h;          // fp8 image of h [L, I] in the bf16 h buffer; its scales in the xn buffer (L * dm * 2 bytes >= L * 4 * nkp(I))
  char* const hq_mx = ws.xn;
  const bool h_fused = fused_q && (int64_t)ttvk_mx_scale_ld(I) <= (int64_t)dm * 2;
  if (h_fused) { f.yq = hq; f.yq_mx = hq_mx; }
  TTV_TRY(ttvk_gemm_fp8(EPI_GEGLU, f, ws.rstd, lw.w12_f8_scale, s, ws.f8mx, lw.w12_mx));
  GemmArgs f3 = {};
  if (h_fused) {
    f3.x = hq;
  } else {
    TTV_TRY(ttvk_quant_mx_fp8(ws.h, dt, I, ws.f8, I, ws.f8mx, nullptr, L, I, s));
    f3.x = ws.f8;
  }
  f3.dtype = dt; f3.ldx = I; f3.w = lw.w3_f8; f3.ldw = I; f3.M = L; f3.N = dm; f3.K = I; f3.resid = ws.x; f3.ldr = dm;
  f3.alpha = i == 0 ? 1.f : d->alpha; f3.y = ws.x; f3.ldy = dm;
  TTV_TRY(ttvk_gemm_fp8(EPI_RESID_T, f3, nullptr, lw.w3_f8_scale, s, h_fused ? hq_mx : ws.f8mx, lw.w3_mx));
  rstd_valid = false;
  if (i > 0) {
    // the next layer's to_qkv operand rides along (the caller only uses it when the next layer runs this path too)
    TTV_TRY(ttvk_rmsnorm(ws.x, dt, dm, nullptr, ws.x, dt, dm, nullptr, lw.ffd_post_ln, L, dm, d->eps, s, ws.rstd, fused_q ? ws.f8 : nullptr,
                         fused_q ? ws.f8mx : nullptr));
    rstd_valid = true;
    xq_valid = fused_q;
  }
  return TTV_OK;
}

// Does a linear of the generic layer path run on the row-scaled e4m3 image (round 2's mixed bf16 / fp8 mode)?  Only when the image IS
// row-scaled: `mx` set means the pointers hold the block-scaled image of W * gain, which the row-scaled GEMM would mis-read (gain applied
// twice, block scales dropped).
static bool rows_f8(int dt, int dm, const void* img, const void* row_scale, const void* mx, const void* folded) {
  return dt == TTV_BF16 && img && row_scale && !mx && dm % 128 == 0 && !(dm == 256 && folded);
}

// One ResidualAttentionBlock stack (reference transformer.py:126-146) on ws.x in place.
static int run_layers(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_batch* b, const TowerWs& ws, hipStream_t s) {
  const int L = b->total_rows, dm = d->width, g = d->kv_heads * d->head_dim, dt = d->dtype;
  const int split3 = (dt == TTV_F32 && w->f32_split3) ? 1 : 0;       // fp32 towers on the three-pass bf16 kernels (ttv_tower_weights.f32_split3)
  // pre-norm gains folded into the weight (to_qkv_pn / w12_pn) with the row statistic applied to the GEMM's output rows: the wide bf16
  // towers, and the split-bf16 towers of any width (their weights are repacked anyway; the exact-fp32 towers keep the reference's order)
  const bool gen_ok = (dt == TTV_BF16 && dm != 256) || split3;
  static const bool s3img_env = !(getenv("TTV_SPLIT3_IMAGES") && getenv("TTV_SPLIT3_IMAGES")[0] == '0');
  const int s3img = (split3 && s3img_env && !(g_ttv_debug & 4096)) ? 1 : 0;     // ttv_debug_set bit 12: fp32 activations, split inside the GEMMs (tests)
  const int nq = 2 * dm + 2 * g;
  bool qkv_ready = false;   // the previous layer's tail kernel already produced this layer's rotated qkv
  bool rstd_valid = false;  // ws.rstd holds rsqrt(mean(x^2) + eps) of the current ws.x (written by the kernel that produced x)
  bool xq_valid = false;    // ws.f8 / ws.f8mx hold the block-scaled e4m3 image of the current ws.x (run_layer_mx)
  static const bool attn_pipe = getenv("TTV_ATTN_PIPE") && getenv("TTV_ATTN_PIPE")[0] == '1';   // opt-in pipelined attention kernel
  // The encoder's output is its latent rows (blocks.py:101-103): with the batch's latent-query table the LAST layer runs its attention for
  // those query rows only and everything behind the attention on the sum K_b latent rows, gathered into compact buffers (ws.xl, ws.aol) and
  // scattered back into ws.x at the end.  Row-wise kernels on other rows: the values of the rows that are read are the same bits.
  // TTV_ENC_LATENT_LAST=0: every row, as the reference computes it (A/B, tests).  Not for the block-scaled fp8 layers (run_layer_mx).
  static const bool lat_env = !(getenv("TTV_ENC_LATENT_LAST") && getenv("TTV_ENC_LATENT_LAST")[0] == '0');
  const bool lat_last = lat_env && !(g_ttv_debug & 524288) && d->kind == TTV_ENCODER && b->qblocks_latent && b->n_qblocks_latent > 0 &&
                        b->latent_rows && b->sum_tokens > 0 && b->sum_tokens < L;
  // The decoder's output is its patch rows (blocks.py:171): with the batch's patch-query table the LAST layer's attention skips the query
  // blocks that hold latent rows only.  Their rows of ws.ao keep the previous layer's values (finite), everything behind the attention
  // is row-wise, the tail gathers patch rows: no patch row changes a bit.  TTV_DEC_PATCH_LAST=0 / debug bit 21: every block (A/B, tests).
  static const bool pat_env = !(getenv("TTV_DEC_PATCH_LAST") && getenv("TTV_DEC_PATCH_LAST")[0] == '0');
  const bool pat_last = pat_env && !(g_ttv_debug & 2097152) && d->kind == TTV_DECODER && d->layers >= 2 && b->qblocks_patch &&
                        b->n_qblocks_patch > 0 && !split3 && !b->qblocks_paired;
  bool compacted = false;
  for (int i = 0; i < d->layers; ++i) {
    const ttv_layer_weights& lw = w->layers[i];
    static const bool keel_f32 = getenv("TTV_KEEL_F32SUM") && getenv("TTV_KEEL_F32SUM")[0] == '1';
    if (dt == TTV_BF16 && dm != 256 && dm % 128 == 0 && d->inner % 128 == 0 && !keel_f32 && lw.to_qkv_f8 && lw.to_qkv_mx && lw.w12_f8 && lw.w12_mx &&
        lw.out_proj_f8 && lw.out_proj_mx && lw.w3_f8 && lw.w3_mx &&
        !(lat_last && i == d->layers - 1)) {    // the encoder's last layer: its latent rows on the bf16 kernels instead (a ninth of the rows)
      TTV_TRY(run_layer_mx(d, lw, b, ws, i, rstd_valid, xq_valid, attn_pipe, s));
      qkv_ready = false;
      continue;
    }
    xq_valid = false;
    // ---- attention sub-layer (transformer.py:85-104) ----
    // mixed bf16 / fp8 (config #5): the pre-norm output is quantised to e4m3 per token (into the xn buffer: L x dm bytes of values,
    // then L fp32 scales) and the projection runs on the fp8 MFMA; everything downstream is unchanged
    // NOT when the images are the block-scaled ones (to_qkv_mx / w12_mx set: e4m3 of W * gain divided by the row factor AND the per-32
    // E8M0 scales, which only run_layer_mx's GEMMs undo): a layer of an MX tower that lands here - the encoder's latent-only last layer,
    // every layer under TTV_KEEL_F32SUM=1 - runs its projections on the bf16 kernels from the folded weights
    const bool f8_qkv = rows_f8(dt, dm, lw.to_qkv_f8, lw.to_qkv_f8_scale, lw.to_qkv_mx, lw.to_qkv_pn);
    const bool f8_w12 = rows_f8(dt, dm, lw.w12_f8, lw.w12_f8_scale, lw.w12_mx, lw.w12_pn);
    float* const f8_scales = reinterpret_cast<float*>(ws.xn + (((size_t)L * dm + 255) & ~(size_t)255));
    if (!qkv_ready && f8_qkv) {
      TTV_TRY(ttvk_quant_rows_fp8(ws.x, dt, dm, lw.pre_ln, d->eps, ws.xn, dm, f8_scales, L, dm, s));
      GemmArgs a = {};
      a.dtype = dt; a.x = ws.xn; a.ldx = dm; a.w = lw.to_qkv_f8; a.ldw = dm; a.M = L; a.N = nq; a.K = dm; a.y = ws.qkv; a.ldy = nq;
      a.rope_cs = b->rope_cs; a.rope_q_end = dm; a.rope_k_begin = 2 * dm; a.rope_k_end = 2 * dm + g;
      TTV_TRY(ttvk_gemm_fp8(EPI_QKV_ROPE, a, f8_scales, lw.to_qkv_f8_scale, s));
    } else if (!qkv_ready) {
      const bool fold_qkv = dt == TTV_BF16 && dm == 256 && lw.to_qkv_pn;
      // other widths: the gain is folded into the weight as well (to_qkv_pn), the row statistic comes from the kernel that
      // produced x (the KEEL post-norm below writes it) or from one light pass, and multiplies the GEMM's output rows - no
      // stand-alone RMSNorm launch, no normalised copy of x
      const bool fold_gen = gen_ok && lw.to_qkv_pn;
      if (fold_gen && !rstd_valid) TTV_TRY(ttvk_row_rstd(ws.x, dt, dm, ws.rstd, L, dm, d->eps, s));
      // split-bf16 towers: the normalised row is written as the projection's split image (its producer splits it once, the GEMM's
      // staging threads copy bytes); TTV_SPLIT3_IMAGES=0 keeps fp32 activations and the split in the GEMM (A/B)
      if (!fold_qkv && !fold_gen) TTV_TRY(ttvk_rmsnorm(ws.x, dt, dm, nullptr, ws.xn, dt, dm, nullptr, lw.pre_ln, L, dm, d->eps, s, nullptr, nullptr, nullptr, s3img));
      GemmArgs a = {};
      a.dtype = dt; a.split3 = split3; a.x_image = s3img && !fold_qkv && !fold_gen; a.y_image = s3img ? 2 : 0;
      a.prenorm = fold_qkv; a.eps = d->eps;
      a.row_scale = fold_gen ? ws.rstd : nullptr;
      const void* w_plain = (dt == TTV_BF16 && lw.to_qkv_qs) ? lw.to_qkv_qs : lw.to_qkv;   // inference copy with scaled q rows, if packed
      a.x = (fold_qkv || fold_gen) ? ws.x : ws.xn; a.ldx = dm; a.w = (fold_qkv || fold_gen) ? lw.to_qkv_pn : w_plain; a.ldw = dm; a.M = L; a.N = nq; a.K = dm; a.y = ws.qkv; a.ldy = nq;
      a.rope_cs = b->rope_cs; a.rope_q_end = dm; a.rope_k_begin = 2 * dm; a.rope_k_end = 2 * dm + g;
      a.rope_ids = b->rope_ids; a.rope_base = b->rope_ids ? b->rope_base : nullptr;
      TTV_TRY(ttvk_gemm(EPI_QKV_ROPE, a, s));
    }
    // q arrives pre-scaled when the projection used the folded weight whose q rows carry scale * log2(e)
    const bool q_scaled = dt == TTV_BF16 && (lw.to_qkv_pn ? lw.qkv_q_prescaled != 0 : lw.to_qkv_qs != nullptr);
    rstd_valid = false;
    qkv_ready = false;
    const bool lat_now = lat_last && i == d->layers - 1;
    if (lat_now)
      // (the latent table holds full items only; it is declared so only when the batch's own table is too: ttvk_attention picks its
      // kernel by that flag, and the latent-rows forward must run the kernel the all-rows forward runs - same bits, tested)
      TTV_TRY(ttvk_attention(ws.qkv, nq, ws.ao, dm, b->cu_seqlens, b->qblocks_latent, b->n_qblocks_latent, d->q_heads, d->kv_heads, d->head_dim,
                             TTV_ATTN_GATE | (b->qblocks_all_full ? TTV_ATTN_ALLFULL : 0) | (q_scaled ? TTV_ATTN_QSCALED : 0) | (split3 ? TTV_ATTN_SPLIT3 : 0) |
                                 (s3img ? (TTV_ATTN_SPLIT_OUT | TTV_ATTN_SPLIT_IN) : 0), dt, s));
    else if (pat_last && i == d->layers - 1 && dt == TTV_BF16 && !attn_pipe)
      TTV_TRY(ttvk_attention(ws.qkv, nq, ws.ao, dm, b->cu_seqlens, b->qblocks_patch, b->n_qblocks_patch, d->q_heads, d->kv_heads, d->head_dim,
                             TTV_ATTN_GATE | (b->qblocks_all_full ? TTV_ATTN_ALLFULL : 0) | (q_scaled ? TTV_ATTN_QSCALED : 0), dt, s));
    else if (q_scaled && b->items64 && b->n_items64 > 0 && d->head_dim == 64)
      TTV_TRY(ttvk_attention64(ws.qkv, nq, ws.ao, dm, b->cu_seqlens, b->items64, b->n_items64, d->q_heads, d->kv_heads,
                               TTV_ATTN_GATE | TTV_ATTN_QSCALED, s));
    else
    TTV_TRY(ttvk_attention(ws.qkv, nq, ws.ao, dm, b->cu_seqlens, b->qblocks, b->n_qblocks, d->q_heads, d->kv_heads, d->head_dim,
                           TTV_ATTN_GATE | (b->qblocks_paired ? TTV_ATTN_PAIRED : 0) | (q_scaled ? TTV_ATTN_QSCALED : 0) |
                               (b->qblocks_all_full ? TTV_ATTN_ALLFULL : 0) | (attn_pipe ? TTV_ATTN_PIPE : 0) | (split3 ? TTV_ATTN_SPLIT3 : 0) |
                               (s3img ? (TTV_ATTN_SPLIT_OUT | TTV_ATTN_SPLIT_IN) : 0), dt, s));
    // from here on the layer works on (cx, cao, Lc): the whole packed batch, or - last encoder layer - its latent rows, compact
    char* cx = ws.x;
    char* cao = ws.ao;
    int Lc = L;
    if (lat_now) {
      const int64_t rb = (int64_t)dm * esize(dt);       // a split image (hi0..3 | lo0..3 per 16 bytes) has the bytes of the fp32 row
      TTV_TRY(ttvk_copy_rows(ws.x, (int64_t)dm * esize(dt), b->latent_rows, ws.xl, (int64_t)dm * esize(dt), nullptr, b->sum_tokens, dm * (int)esize(dt), s));
      TTV_TRY(ttvk_copy_rows(ws.ao, rb, b->latent_rows, ws.aol, rb, nullptr, b->sum_tokens, (int)rb, s));
      cx = ws.xl; cao = ws.aol; Lc = b->sum_tokens;
      compacted = true;
    }
    // TTV_FUSED_MLP=0 selects the unfused kernel sequence (A/B measurements; same results up to bf16 rounding of h).
    // TTV_FUSED_QKV=1 additionally folds the NEXT layer's QKV projection + rotary into the tail kernel: correct and tested,
    // but measured 3 % slower end to end than the stand-alone QKV kernel (the phase runs on the 192 CUs / uneven wave pairs
    // of the tail kernel: 31 us against 35 us stand-alone in isolation, worse in the pipeline), so it is opt-in.
    static const bool keel_f32sum = getenv("TTV_KEEL_F32SUM") && getenv("TTV_KEEL_F32SUM")[0] == '1';
    static const bool use_fused_mlp = !(getenv("TTV_FUSED_MLP") && getenv("TTV_FUSED_MLP")[0] == '0');
    static const bool use_fused_qkv = getenv("TTV_FUSED_QKV") && getenv("TTV_FUSED_QKV")[0] == '1';
    if (use_fused_mlp && ttvk_mlp_fused_supported(dt, dm, d->inner) && lw.mlp_pack) {
      // one kernel for the rest of the layer: out_proj + residual/KEEL norm, then pre-norm + w12 + GEGLU + w3 +
      // residual/KEEL norm, in place on x, and (when the pack carries it) the NEXT layer's pre_ln + to_qkv + rotary
      MlpNextQkv nx = {};
      const bool back = use_fused_qkv && i + 1 < d->layers && lw.mlp_pack_qkv_rows == nq && nq % 64 == 0 && dm % 64 == 0 && g % 64 == 0;
      if (back) { nx.qkv = ws.qkv; nx.ld = nq; nx.rope_cs = b->rope_cs; nx.rows = nq; nx.rope_q_end = dm; nx.rope_k_begin = 2 * dm; nx.rope_k_end = 2 * dm + g; }
      TTV_TRY(ttvk_mlp_fused(cao, dm, i == 0 ? nullptr : lw.attn_post_ln, i == 0 ? 1.f : d->alpha, cx, dm, lw.mlp_pack, d->inner,
                             cx, dm, i == 0 ? nullptr : lw.ffd_post_ln, i == 0 ? 1.f : d->alpha, d->eps, Lc, back ? &nx : nullptr, s));
      qkv_ready = back;
      continue;
    }
    GemmArgs o = {};
    o.dtype = dt; o.split3 = split3; o.x_image = s3img;
    o.x = cao; o.ldx = dm; o.w = lw.out_proj; o.ldw = dm; o.M = Lc; o.N = dm; o.K = dm; o.resid = cx; o.ldr = dm;
    if (i == 0) {
      o.alpha = 1.f; o.y = cx; o.ldy = dm;
      TTV_TRY(ttvk_gemm(EPI_RESID_T, o, s));
    } else if (ttvk_gemm_supports_resid_norm(dt, dm, dm)) {
      // x <- RMSNorm(alpha*x + ao@Wo^T) * gain in one kernel; in place on x is safe: a token row is read (as residual)
      // and written by the same wave only
      o.alpha = d->alpha; o.y = cx; o.ldy = dm; o.norm_gain = lw.attn_post_ln; o.eps = d->eps;
      TTV_TRY(ttvk_gemm(EPI_RESID_NORM, o, s));
    } else {
      // wide towers: the KEEL sum alpha * x + f(x) leaves the GEMM through HBM and a row kernel normalises it.  bf16 towers store it
      // in the compute dtype, in place on x (a token row element is read as residual and written by the same lane) - the reference's
      // autocast rounds that sum to bf16 as well (transformer.py:141: bf16 * alpha + bf16) -: half the bytes of the fp32 buffer on
      // both kernels.  fp32 towers, and TTV_KEEL_F32SUM=1 (A/B), keep the fp32 buffer.
      const bool want = gen_ok && lw.w12_pn && !f8_w12;
      if (dt == TTV_BF16 && !keel_f32sum) {
        o.alpha = d->alpha; o.y = cx; o.ldy = dm;
        TTV_TRY(ttvk_gemm(EPI_RESID_T, o, s));
        TTV_TRY(ttvk_rmsnorm(cx, dt, dm, nullptr, cx, dt, dm, nullptr, lw.attn_post_ln  , Lc, dm, d->eps, s, want ? ws.rstd : nullptr));
      } else {
        o.alpha = d->alpha; o.y = ws.y32; o.ldy = dm;
        TTV_TRY(ttvk_gemm(EPI_RESID_F32, o, s));
        TTV_TRY(ttvk_rmsnorm(ws.y32, TTV_F32, dm, nullptr, cx, dt, dm, nullptr, lw.attn_post_ln  , Lc, dm, d->eps, s, want ? ws.rstd : nullptr));
      }
      rstd_valid = want;
    }
    // ---- GEGLU sub-layer (transformer.py:47-56) ----
    const bool fold_ffd = dt == TTV_BF16 && dm == 256 && lw.w12_pn;
    const bool fold_ffd_gen = gen_ok && lw.w12_pn && !f8_w12;
    if (f8_w12) {
      TTV_TRY(ttvk_quant_rows_fp8(cx, dt, dm, lw.ffd_norm, d->eps, ws.xn, dm, f8_scales, Lc, dm, s));
      GemmArgs f = {};
      f.dtype = dt; f.x = ws.xn; f.ldx = dm; f.w = lw.w12_f8; f.ldw = dm; f.M = Lc; f.N = d->inner; f.K = dm; f.y = ws.h; f.ldy = d->inner;
      TTV_TRY(ttvk_gemm_fp8(EPI_GEGLU, f, f8_scales, lw.w12_f8_scale, s));
    } else {
    if (fold_ffd_gen && !rstd_valid) TTV_TRY(ttvk_row_rstd(cx, dt, dm, ws.rstd  , Lc, dm, d->eps, s));
    if (!fold_ffd && !fold_ffd_gen) TTV_TRY(ttvk_rmsnorm(cx, dt, dm, nullptr, ws.xn, dt, dm, nullptr, lw.ffd_norm  , Lc, dm, d->eps, s, nullptr, nullptr, nullptr, s3img));
    GemmArgs f = {};
    f.dtype = dt; f.split3 = split3; f.x_image = s3img && !fold_ffd && !fold_ffd_gen; f.y_image = s3img;
    f.prenorm = fold_ffd; f.eps = d->eps;
    f.row_scale = fold_ffd_gen ? ws.rstd : nullptr;
    f.x = (fold_ffd || fold_ffd_gen) ? cx : ws.xn; f.ldx = dm; f.w = (fold_ffd || fold_ffd_gen) ? lw.w12_pn : lw.w12; f.ldw = dm; f.M = Lc; f.N = d->inner; f.K = dm; f.y = ws.h; f.ldy = d->inner;
    TTV_TRY(ttvk_gemm(EPI_GEGLU, f, s));
    }
    rstd_valid = false;
    GemmArgs f3 = {};
    f3.dtype = dt; f3.split3 = split3; f3.x_image = s3img;
    f3.x = ws.h; f3.ldx = d->inner; f3.w = lw.w3; f3.ldw = d->inner; f3.M = Lc; f3.N = dm; f3.K = d->inner; f3.resid = cx; f3.ldr = dm;
    if (i == 0) {
      f3.alpha = 1.f; f3.y = cx; f3.ldy = dm;
      TTV_TRY(ttvk_gemm(EPI_RESID_T, f3, s));
    } else if (ttvk_gemm_supports_resid_norm(dt, dm, d->inner)) {
      // x <- RMSNorm(alpha*x + h@W3^T) * gain in one full-row kernel (in place: a token row is read and written by one block)
      f3.alpha = d->alpha; f3.y = cx; f3.ldy = dm; f3.norm_gain = lw.ffd_post_ln; f3.eps = d->eps;
      TTV_TRY(ttvk_gemm(EPI_RESID_NORM, f3, s));
    } else {
      const bool want = gen_ok && i + 1 < d->layers && w->layers[i + 1].to_qkv_pn &&
                        !rows_f8(dt, dm, w->layers[i + 1].to_qkv_f8, w->layers[i + 1].to_qkv_f8_scale, w->layers[i + 1].to_qkv_mx, w->layers[i + 1].to_qkv_pn);
      if (dt == TTV_BF16 && !keel_f32sum) {      // see the attention sub-layer above
        f3.alpha = d->alpha; f3.y = cx; f3.ldy = dm;
        TTV_TRY(ttvk_gemm(EPI_RESID_T, f3, s));
        TTV_TRY(ttvk_rmsnorm(cx, dt, dm, nullptr, cx, dt, dm, nullptr, lw.ffd_post_ln  , Lc, dm, d->eps, s, want ? ws.rstd : nullptr));
      } else {
        f3.alpha = d->alpha; f3.y = ws.y32; f3.ldy = dm;
        TTV_TRY(ttvk_gemm(EPI_RESID_F32, f3, s));
        TTV_TRY(ttvk_rmsnorm(ws.y32, TTV_F32, dm, nullptr, cx, dt, dm, nullptr, lw.ffd_post_ln  , Lc, dm, d->eps, s, want ? ws.rstd : nullptr));
      }
      rstd_valid = want;
    }
  }
  if (compacted)     // the latent rows back where the encoder's tail (and anybody else) reads them
    TTV_TRY(ttvk_copy_rows(ws.xl, (int64_t)dm * esize(dt), nullptr, ws.x, (int64_t)dm * esize(dt), b->latent_rows, b->sum_tokens, dm * (int)esize(dt), s));
  return TTV_OK;
}

extern "C" {

const char* ttv_error_string(void) { return g_err; }
int ttv_version(void) { return 100; }

int ttv_fsq_forward(const ttv_fsq_params* p, const void* z, int z_dtype, int rows, void* codes, int codes_dtype, int32_t* indices,
                    float* bounded, void* stream) {
  TTV_CHECK_ARG(rows >= 0 && (rows == 0 || (z && codes && indices)), "fsq_forward: null buffer");
  return ttvk_fsq_forward(p, z, z_dtype, rows, codes, codes_dtype, indices, bounded, (hipStream_t)stream);
}

int ttv_fsq_indices_to_codes(const ttv_fsq_params* p, const int32_t* indices, int rows, void* codes, int codes_dtype, void* stream) {
  TTV_CHECK_ARG(rows >= 0 && (rows == 0 || (indices && codes)), "fsq_indices_to_codes: null buffer");
  return ttvk_fsq_indices_to_codes(p, indices, rows, codes, codes_dtype, (hipStream_t)stream);
}

int ttv_vq_codebook_norms(const void* codebook, int dtype, int ld, int N, int C, float* cnorm, void* stream) {
  TTV_CHECK_ARG(N == 0 || (codebook && cnorm), "vq_codebook_norms: null buffer");
  return ttvk_vq_norms(codebook, dtype, ld, N, C, cnorm, (hipStream_t)stream);
}

int64_t ttv_vq_workspace_bytes(int rows) { return ttvk_vq_workspace_bytes(rows); }

int ttv_vq_l2_argmin(const void* z, int dtype, int ldz, const void* codebook, int ldc, const float* cnorm, int rows, int N, int C,
                     int32_t* indices, float* best_dist, void* workspace, int64_t workspace_bytes, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (z && codebook && cnorm && indices), "vq_l2_argmin: null buffer");
  TTV_CHECK_ARG(ldz >= C && ldc >= C, "vq_l2_argmin: leading dims smaller than the codebook dim");
  return ttvk_vq_l2_argmin(z, dtype, ldz, codebook, ldc, cnorm, rows, N, C, indices, best_dist, workspace, workspace_bytes, (hipStream_t)stream);
}

int ttv_vq_lookup(const void* codebook, int dtype, int ldc, const int32_t* indices, int rows, int C, void* codes, int ldo, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (codebook && indices && codes), "vq_lookup: null buffer");
  return ttvk_vq_lookup(codebook, dtype, ldc, indices, rows, C, codes, ldo, (hipStream_t)stream);
}

int ttv_vq_lookup_backward(const void* dcodes, int dtype, int ld, const int32_t* indices, int rows, int C, float* dcodebook, int ldc, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (dcodes && indices && dcodebook), "vq_lookup_backward: null buffer");
  TTV_CHECK_ARG(ld >= C && ldc >= C, "vq_lookup_backward: leading dims smaller than the codebook dim");
  return ttvk_vq_lookup_bwd(dcodes, dtype, ld, indices, rows, C, dcodebook, ldc, (hipStream_t)stream);
}

int ttv_quant_rows_fp8(const void* in, int dtype, int ld_in, const float* gain, float eps, void* out, int ld_out, float* scales, int rows,
                       int width, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (in && out && scales), "quant_rows_fp8: null buffer");
  return ttvk_quant_rows_fp8(in, dtype, ld_in, gain, eps, out, ld_out, scales, rows, width, (hipStream_t)stream);
}

int ttv_split3_pack(const float* w, int ldw, void* out, int ldo, int rows, int K, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (w && out), "split3_pack: null buffer");
  return ttvk_split3_pack(w, ldw, out, ldo, rows, K, (hipStream_t)stream);
}

int ttv_linear_split3(const float* x, int ldx, const void* w_image, int ldw, const float* bias, float* y, int ldy, int M, int N, int K, void* stream) {
  TTV_CHECK_ARG(M == 0 || (x && w_image && y), "linear_split3: null buffer");
  GemmArgs a = {};
  a.dtype = TTV_F32; a.split3 = 1; a.x = x; a.ldx = ldx; a.w = w_image; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.y = y; a.ldy = ldy; a.bias = bias;
  return ttvk_gemm(EPI_STORE, a, (hipStream_t)stream);
}

int64_t ttv_mx_scale_bytes_per_row(int width) { return ttvk_mx_scale_ld(width); }

int ttv_quant_mx_fp8(const void* in, int dtype, int ld_in, void* out, int ld_out, void* mx, float* row_scales, int rows, int width, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (in && out && mx), "quant_mx_fp8: null buffer");
  TTV_CHECK_ARG(ld_in >= width && ld_out >= width, "quant_mx_fp8: leading dims smaller than the width");
  return ttvk_quant_mx_fp8(in, dtype, ld_in, out, ld_out, mx, row_scales, rows, width, (hipStream_t)stream);
}

int ttv_linear_fp8_mx(const void* xq, int ldx, const void* x_mx, const float* x_row_scale, const void* wq, int ldw, const void* w_mx,
                      const float* w_row_scale, void* y, int ldy, int M, int N, int K, int epilogue, const float* rope_cs, int d_model,
                      int gqa_dim, const void* resid, int ldr, float alpha, void* stream) {
  TTV_CHECK_ARG(M == 0 || (xq && wq && y && x_mx && w_mx), "linear_fp8_mx: null buffer");
  TTV_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "linear_fp8_mx: epilogue 0 (store), 1 (qkv + rotary), 2 (GEGLU) or 3 (alpha * resid + acc)");
  GemmArgs a = {};
  a.dtype = TTV_BF16; a.x = xq; a.ldx = ldx; a.w = wq; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.y = y; a.ldy = ldy;
  if (epilogue == 1) {
    TTV_CHECK_ARG(rope_cs && N == 2 * d_model + 2 * gqa_dim, "linear_fp8_mx: qkv epilogue needs rope_cs and N = 2 d_model + 2 gqa_dim");
    a.rope_cs = rope_cs; a.rope_q_end = d_model; a.rope_k_begin = 2 * d_model; a.rope_k_end = 2 * d_model + gqa_dim;
  }
  if (epilogue == 3) { a.resid = resid; a.ldr = ldr; a.alpha = alpha; }
  return ttvk_gemm_fp8(epilogue == 0 ? EPI_STORE : epilogue == 1 ? EPI_QKV_ROPE : epilogue == 2 ? EPI_GEGLU : EPI_RESID_T, a, x_row_scale,
                       w_row_scale, (hipStream_t)stream, x_mx, w_mx);
}

int ttv_linear_fp8(const void* xq, int ldx, const float* x_scale, const void* wq, int ldw, const float* w_scale, void* y, int ldy, int M, int N,
                   int K, int epilogue, const float* rope_cs, int d_model, int gqa_dim, void* stream) {
  TTV_CHECK_ARG(M == 0 || (xq && wq && y), "linear_fp8: null buffer");
  TTV_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "linear_fp8: epilogue 0 (store), 1 (qkv + rotary) or 2 (GEGLU)");
  GemmArgs a = {};
  a.dtype = TTV_BF16; a.x = xq; a.ldx = ldx; a.w = wq; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.y = y; a.ldy = ldy;
  if (epilogue == 1) {
    TTV_CHECK_ARG(rope_cs && N == 2 * d_model + 2 * gqa_dim, "linear_fp8: qkv epilogue needs rope_cs and N = 2 d_model + 2 gqa_dim");
    a.rope_cs = rope_cs; a.rope_q_end = d_model; a.rope_k_begin = 2 * d_model; a.rope_k_end = 2 * d_model + gqa_dim;
  }
  return ttvk_gemm_fp8(epilogue == 0 ? EPI_STORE : epilogue == 1 ? EPI_QKV_ROPE : EPI_GEGLU, a, x_scale, w_scale, (hipStream_t)stream);
}

int ttv_rmsnorm(const void* in, int in_dtype, int ld_in, const int32_t* src_rows, void* out, int out_dtype, int ld_out,
                const int32_t* dst_rows, const float* gain, int rows, int width, float eps, void* stream) {
  TTV_CHECK_ARG(rows >= 0 && (rows == 0 || (in && out && gain)), "rmsnorm: null buffer");
  return ttvk_rmsnorm(in, in_dtype, ld_in, src_rows, out, out_dtype, ld_out, dst_rows, gain, rows, width, eps, (hipStream_t)stream);
}

int ttv_rope_apply(void* x, int dtype, int ld, int rows, int heads, const float* rope_cs, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (x && rope_cs), "rope_apply: null buffer");
  TTV_CHECK_ARG(ld % 4 == 0 && ld >= heads * 64, "rope_apply: bad leading dim");
  return ttvk_rope_apply(x, dtype, ld, rows, heads, rope_cs, (hipStream_t)stream);
}

int ttv_linear(const void* x, int ldx, const void* w, int ldw, const void* bias, const float* add_scalar, void* y, int ldy, int M,
               int N, int K, int dtype, void* stream) {
  TTV_CHECK_ARG(M == 0 || (x && w && y), "linear: null buffer");
  GemmArgs a = {};
  a.dtype = dtype; a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.y = y; a.ldy = ldy;
  a.bias = bias; a.add_scalar = add_scalar;
  return ttvk_gemm(EPI_STORE, a, (hipStream_t)stream);
}

int ttv_linear_qkv_rope(const void* x, int ldx, const void* w, int ldw, void* y, int ldy, int M, int d_model, int gqa_dim,
                        const float* rope_cs, int dtype, void* stream) {
  TTV_CHECK_ARG(M == 0 || (x && w && y && rope_cs), "linear_qkv_rope: null buffer");
  GemmArgs a = {};
  a.dtype = dtype; a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.M = M; a.N = 2 * d_model + 2 * gqa_dim; a.K = d_model;
  a.y = y; a.ldy = ldy; a.rope_cs = rope_cs; a.rope_q_end = d_model; a.rope_k_begin = 2 * d_model; a.rope_k_end = 2 * d_model + gqa_dim;
  return ttvk_gemm(EPI_QKV_ROPE, a, (hipStream_t)stream);
}

int ttv_linear_geglu(const void* x, int ldx, const void* w, int ldw, void* y, int ldy, int M, int I, int K, int dtype, void* stream) {
  TTV_CHECK_ARG(M == 0 || (x && w && y), "linear_geglu: null buffer");
  GemmArgs a = {};
  a.dtype = dtype; a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.M = M; a.N = I; a.K = K; a.y = y; a.ldy = ldy;
  return ttvk_gemm(EPI_GEGLU, a, (hipStream_t)stream);
}

int ttv_linear_residual(const void* x, int ldx, const void* w, int ldw, const void* resid, int ldr, float alpha, void* y, int ldy,
                        int y_f32, int M, int N, int K, int dtype, void* stream) {
  TTV_CHECK_ARG(M == 0 || (x && w && y && resid), "linear_residual: null buffer");
  GemmArgs a = {};
  a.dtype = dtype; a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.y = y; a.ldy = ldy;
  a.resid = resid; a.ldr = ldr; a.alpha = alpha;
  return ttvk_gemm(y_f32 ? EPI_RESID_F32 : EPI_RESID_T, a, (hipStream_t)stream);
}

int ttv_linear_residual_norm(const void* x, int ldx, const void* w, int ldw, const void* resid, int ldr, float alpha,
                             const float* gain, float eps, void* y, int ldy, int M, int N, int K, int dtype, void* stream) {
  TTV_CHECK_ARG(M == 0 || (x && w && y && resid && gain), "linear_residual_norm: null buffer");
  GemmArgs a = {};
  a.dtype = dtype; a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.y = y; a.ldy = ldy;
  a.resid = resid; a.ldr = ldr; a.alpha = alpha; a.norm_gain = gain; a.eps = eps;
  if (!ttvk_gemm_supports_resid_norm(dtype, N, K)) {
    ttv_set_error("linear_residual_norm: only bf16 with N == 256 has a fused kernel");
    return TTV_ERR_UNSUPPORTED;
  }
  return ttvk_gemm(EPI_RESID_NORM, a, (hipStream_t)stream);
}

int64_t ttv_mlp_pack_bytes(int inner, int next_qkv_rows) { return ttvk_mlp_pack_bytes(inner, next_qkv_rows); }

int ttv_mlp_pack(const void* w12_folded, const void* w3, const void* out_proj, const void* next_qkv_folded, int next_qkv_rows, int inner,
                 int width, int dtype, void* packed, void* stream) {
  if (!ttvk_mlp_fused_supported(dtype, width, inner)) {
    ttv_set_error("mlp_pack: only bf16, width 256, inner %% 32 == 0");
    return TTV_ERR_UNSUPPORTED;
  }
  return ttvk_mlp_pack(w12_folded, w3, out_proj, next_qkv_folded, next_qkv_rows, inner, packed, (hipStream_t)stream);
}

int ttv_mlp_fused(const void* x, int ldx, const void* mlp_packed, int inner, void* y, int ldy, const float* post_gain, float alpha,
                  float eps, int M, int width, int dtype, void* stream) {
  if (!ttvk_mlp_fused_supported(dtype, width, inner)) {
    ttv_set_error("mlp_fused: only bf16, width 256, inner %% 32 == 0");
    return TTV_ERR_UNSUPPORTED;
  }
  return ttvk_mlp_fused(nullptr, 0, nullptr, 1.f, x, ldx, mlp_packed, inner, y, ldy, post_gain, alpha, eps, M, nullptr, (hipStream_t)stream);
}

int ttv_layer_tail_fused(const void* ao, int ldao, const float* attn_post_gain, float attn_alpha, const void* x, int ldx,
                         const void* mlp_packed, int inner, void* y, int ldy, const float* ffd_post_gain, float ffd_alpha, float eps,
                         int M, int width, int dtype, const ttv_next_qkv* next, void* stream) {
  if (!ttvk_mlp_fused_supported(dtype, width, inner)) {
    ttv_set_error("layer_tail_fused: only bf16, width 256, inner %% 32 == 0");
    return TTV_ERR_UNSUPPORTED;
  }
  TTV_CHECK_ARG(ao, "layer_tail_fused: null attention output");
  MlpNextQkv nx = {};
  if (next) { nx.qkv = next->qkv; nx.ld = next->ld; nx.rope_cs = next->rope_cs; nx.rows = next->rows; nx.rope_q_end = next->rope_q_end;
              nx.rope_k_begin = next->rope_k_begin; nx.rope_k_end = next->rope_k_end; }
  return ttvk_mlp_fused(ao, ldao, attn_post_gain, attn_alpha, x, ldx, mlp_packed, inner, y, ldy, ffd_post_gain, ffd_alpha, eps, M,
                        next ? &nx : nullptr, (hipStream_t)stream);
}

int ttv_fill_const_rows(void* x, int dtype, int ld, const int32_t* rows_map, int rows, int width, const float* mask_token,
                        const float* gain, float eps, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (x && rows_map && mask_token && gain), "fill_const_rows: null buffer");
  return ttvk_fill_const_rows(x, dtype, ld, rows_map, rows, width, mask_token, gain, eps, (hipStream_t)stream);
}

int ttv_decoder_embed(const void* codes, int token_size, const void* w, const void* bias, const float* mask_token, const float* gain,
                      void* x, int dtype, int ld, const int32_t* rows_map, int rows, int width, float eps, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (codes && w && bias && mask_token && gain && x && rows_map), "decoder_embed: null buffer");
  return ttvk_dec_embed(codes, token_size, w, bias, mask_token, gain, x, dtype, ld, rows_map, rows, width, eps, (hipStream_t)stream);
}

int ttv_attention(const void* qkvg, int ld, void* out, int ldo, const int32_t* cu_seqlens, const int32_t* qblocks, int n_qblocks,
                  int q_heads, int kv_heads, int head_dim, int flags, int dtype, void* stream) {
  TTV_CHECK_ARG(n_qblocks == 0 || (qkvg && out && cu_seqlens && qblocks), "attention: null buffer");
  return ttvk_attention(qkvg, ld, out, ldo, cu_seqlens, qblocks, n_qblocks, q_heads, kv_heads, head_dim, flags, dtype, (hipStream_t)stream);
}
int ttv_attention64(const void* qkvg, int ld, void* out, int ldo, const int32_t* cu_seqlens, const int32_t* items, int n_items, int q_heads,
                    int kv_heads, int head_dim, int flags, int dtype, void* stream) {
  TTV_CHECK_ARG(dtype == TTV_BF16 && head_dim == 64, "attention64: bf16, head_dim 64 only");
  return ttvk_attention64(qkvg, ld, out, ldo, cu_seqlens, items, n_items, q_heads, kv_heads, flags, (hipStream_t)stream);
}

int ttv_patch_gather(const void* const* clips, const int32_t* clip_desc, int clip0, int n_clips, int patch_t, int patch_h, int patch_w,
                     int channels, void* patches, int ld, int dtype, int max_patches_per_clip, void* stream) {
  TTV_CHECK_ARG(n_clips == 0 || (clips && clip_desc && patches), "patch_gather: null buffer");
  return ttvk_patch_copy(false, (void* const*)clips, clip_desc, clip0, n_clips, patch_t, patch_h, patch_w, channels, patches, ld, dtype, max_patches_per_clip, (hipStream_t)stream);
}

int ttv_patch_scatter(const void* patches, int ld, const int32_t* clip_desc, int clip0, int n_clips, int patch_t, int patch_h, int patch_w,
                      int channels, void* const* clips, int dtype, int max_patches_per_clip, void* stream) {
  TTV_CHECK_ARG(n_clips == 0 || (clips && clip_desc && patches), "patch_scatter: null buffer");
  return ttvk_patch_copy(true, clips, clip_desc, clip0, n_clips, patch_t, patch_h, patch_w, channels, (void*)patches, ld, dtype, max_patches_per_clip, (hipStream_t)stream);
}

int64_t ttv_tower_workspace_bytes(const ttv_tower_dims* dims, const ttv_batch* batch) {
  if (check_dims(dims, batch) != TTV_OK) return -1;
  return carve(dims, batch, nullptr).total;
}

int ttv_encoder_forward(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_batch* b, const void* const* clips,
                        const ttv_fsq_params* fsq, float* z, void* codes, int32_t* indices, float* bounded, void* workspace,
                        int64_t workspace_bytes, void* stream) {
  TTV_TRY(check_dims(d, b));
  TTV_CHECK_ARG(d->kind == TTV_ENCODER, "encoder_forward: dims.kind is not TTV_ENCODER");
  TTV_CHECK_ARG(w && w->layers && clips && workspace, "encoder_forward: null argument");
  TTV_CHECK_ARG(fsq || z, "encoder_forward: neither fsq nor z requested");
  TTV_CHECK_ARG(!fsq || (codes && indices), "encoder_forward: fsq needs codes and indices buffers");
  hipStream_t s = (hipStream_t)stream;
  TowerWs ws = carve(d, b, (char*)workspace);
  TTV_CHECK_ARG(ws.total <= workspace_bytes, "encoder_forward: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)ws.total);
  const int dm = d->width, dt = d->dtype, P = b->sum_patches;
  const int pd = d->pix_channels * d->patch_t * d->patch_h * d->patch_w;

  // patchify (utils.py:26-34) + proj_in (blocks.py:91-93); when the shapes allow it the GEMM reads its K = (c, pt, ph, pw)
  // operand straight from the clips (16-byte pixel-row segments) instead of from a gathered [P, pd] copy
  static const bool use_fused_patch = !(getenv("TTV_FUSED_PATCH") && getenv("TTV_FUSED_PATCH")[0] == '0');
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const bool gather = use_fused_patch && dt == TTV_BF16 && d->patch_w == 8 && pow2(d->patch_t) && pow2(d->patch_h) &&
                      b->n_clips <= TTV_MAX_CLIPS_PER_LAUNCH && b->row_seq && pd % 64 == 0 && pd != 256;
  if (!gather) {
    for (int c0 = 0; c0 < b->n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
      const int n = b->n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? b->n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
      TTV_TRY(ttvk_patch_copy(false, (void* const*)(clips + c0), b->clip_desc, c0, n, d->patch_t, d->patch_h, d->patch_w, d->pix_channels, ws.pa, pd, dt, b->max_patches_per_clip, s));
    }
  }
  GemmArgs a = {};
  a.dtype = dt; a.x = ws.pa; a.ldx = pd; a.w = w->proj_in_w; a.ldw = pd; a.M = P; a.N = dm; a.K = pd; a.y = ws.pb; a.ldy = dm;
  a.bias = w->proj_in_b; a.add_scalar = w->mask_token;
  a.split3 = (dt == TTV_F32 && w->f32_split3) ? 1 : 0;
  if (gather) {
    a.gather = 1; a.clips = (void* const*)clips; a.n_clips = b->n_clips; a.clip_desc = b->clip_desc; a.patch_rows = b->patch_rows;
    a.row_seq = b->row_seq; a.patch_t = d->patch_t; a.patch_h = d->patch_h; a.patch_w = d->patch_w;
  }
  TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
  // x[patch rows] = ln_pre_p(patches + mask_token); x[latent rows] = ln_pre_t(mask_token * 1) (blocks.py:95-97)
  TTV_TRY(ttvk_rmsnorm(ws.pb, dt, dm, nullptr, ws.x, dt, dm, b->patch_rows, w->ln_pre_p, P, dm, d->eps, s));
  TTV_TRY(ttvk_fill_const_rows(ws.x, dt, dm, b->latent_rows, b->sum_tokens, dm, w->mask_token, w->ln_pre_t, d->eps, s));

  TTV_TRY(run_layers(d, w, b, ws, s));

  // tokens = proj_out(ln_post(x[latent rows])) -> FSQ (blocks.py:101-103, fsq.py:123-135)
  TTV_TRY(ttvk_enc_tail(ws.x, dt, dm, b->latent_rows, b->sum_tokens, dm, w->ln_post, d->eps, w->proj_out_w, w->proj_out_b, d->token_size, fsq, z, codes, indices, bounded, s));
  return TTV_OK;
}

int ttv_decoder_forward(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_batch* b, const void* codes,
                        void* const* clips_out, void* workspace, int64_t workspace_bytes, void* stream) {
  TTV_TRY(check_dims(d, b));
  TTV_CHECK_ARG(d->kind == TTV_DECODER, "decoder_forward: dims.kind is not TTV_DECODER");
  TTV_CHECK_ARG(w && w->layers && codes && clips_out && workspace, "decoder_forward: null argument");
  hipStream_t s = (hipStream_t)stream;
  TowerWs ws = carve(d, b, (char*)workspace);
  TTV_CHECK_ARG(ws.total <= workspace_bytes, "decoder_forward: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)ws.total);
  const int dm = d->width, dt = d->dtype, P = b->sum_patches;
  const int pd = d->pix_channels * d->patch_t * d->patch_h * d->patch_w;

  // x[latent rows] = ln_pre_t(proj_in(codes) + mask_token); x[patch rows] = ln_pre_p(mask_token * 1) (blocks.py:165-167)
  TTV_TRY(ttvk_dec_embed(codes, d->token_size, w->proj_in_w, w->proj_in_b, w->mask_token, w->ln_pre_t, ws.x, dt, dm, b->latent_rows, b->sum_tokens, dm, d->eps, s));
  TTV_TRY(ttvk_fill_const_rows(ws.x, dt, dm, b->patch_rows, P, dm, w->mask_token, w->ln_pre_p, d->eps, s));

  TTV_TRY(run_layers(d, w, b, ws, s));

  // patches = proj_out(ln_post(x[patch rows])) -> unpatchify (blocks.py:171-176)
  GemmArgs a = {};
  a.dtype = dt; a.w = w->proj_out_w; a.ldw = dm; a.M = P; a.N = pd; a.K = dm; a.y = ws.pa; a.ldy = pd; a.ldx = dm;
  a.split3 = (dt == TTV_F32 && w->f32_split3) ? 1 : 0;
  a.bias = w->proj_out_b;
  if (dt == TTV_BF16 && dm == 256 && w->proj_out_pn && pd % 8 == 0) {
    // ln_post folded into the GEMM: gain in the weight columns, rstd from the register-resident row, rows gathered in place
    a.x = ws.x; a.x_rows = b->patch_rows; a.w = w->proj_out_pn; a.prenorm = 1; a.eps = d->eps;
  } else {
    TTV_TRY(ttvk_rmsnorm(ws.x, dt, dm, b->patch_rows, ws.pb, dt, dm, nullptr, w->ln_post, P, dm, d->eps, s));
    a.x = ws.pb;
  }
  // unpatchify inside the GEMM epilogue (8 consecutive output features = one 16-byte pixel row segment of a patch) when the
  // shapes allow it: saves the [P, pd] round trip and the copy kernel.  TTV_FUSED_PATCH=0 keeps the two-kernel sequence.
  static const bool use_fused_patch = !(getenv("TTV_FUSED_PATCH") && getenv("TTV_FUSED_PATCH")[0] == '0');
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (use_fused_patch && dt == TTV_BF16 && dm == 256 && d->patch_w == 8 && pow2(d->patch_t) && pow2(d->patch_h) &&
      b->n_clips <= TTV_MAX_CLIPS_PER_LAUNCH && b->row_seq && pd % 64 == 0) {
    a.clips = clips_out; a.n_clips = b->n_clips; a.clip_desc = b->clip_desc; a.patch_rows = b->patch_rows; a.row_seq = b->row_seq;
    a.patch_t = d->patch_t; a.patch_h = d->patch_h; a.patch_w = d->patch_w;
    return ttvk_gemm(EPI_STORE_PATCH, a, s);
  }
  TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
  for (int c0 = 0; c0 < b->n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
    const int n = b->n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? b->n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
    TTV_TRY(ttvk_patch_copy(true, clips_out + c0, b->clip_desc, c0, n, d->patch_t, d->patch_h, d->patch_w, d->pix_channels, ws.pa, pd, dt, b->max_patches_per_clip, s));
  }
  return TTV_OK;
}

int ttv_rope_table_build(const float* base_cos, const float* base_sin, int n_ids, int n_freqs, const int32_t* clip_desc, const int32_t* cu_seqlens,
                         const int32_t* row_seq, float* rope_cs, int total_rows, void* stream) {
  TTV_CHECK_ARG(total_rows == 0 || (base_cos && base_sin && clip_desc && cu_seqlens && row_seq && rope_cs), "rope_table_build: null buffer");
  TTV_CHECK_ARG(n_freqs >= 1 && 3 * n_freqs <= 32 && n_ids >= 1, "rope_table_build: bad table shape");
  return ttvk_rope_build(base_cos, base_sin, n_ids, n_freqs, clip_desc, cu_seqlens, row_seq, rope_cs, total_rows, (hipStream_t)stream);
}

int ttv_l1_loss(void* const* recon, void* const* target, void* const* grad, const int32_t* sizes, int n_clips, int dtype, float* loss,
                void* stream) {
  // loss must be zeroed by the caller; clips are processed in groups of TTV_MAX_CLIPS_PER_LAUNCH
  for (int c0 = 0; c0 < n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
    const int n = n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
    TTV_TRY(ttvk_l1_loss(recon + c0, target + c0, grad ? grad + c0 : nullptr, sizes + c0, n, n_clips, dtype, loss, (hipStream_t)stream));
  }
  return TTV_OK;
}

int ttv_clip_from_u8(const void* frames_thwc, int T, int H, int W, void* clip_cthw, int dtype, void* stream) {
  TTV_CHECK_ARG(frames_thwc && clip_cthw && T > 0 && H > 0 && W > 0, "clip_from_u8: bad argument");
  return ttvk_clip_from_u8(frames_thwc, (long long)T * H * W, clip_cthw, dtype, (hipStream_t)stream);
}

int ttv_sq_err_accumulate(void* const* recon, void* const* target, const int32_t* sizes, int n_clips, int dtype, int clamp, double* acc,
                          void* stream) {
  for (int c0 = 0; c0 < n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
    const int n = n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
    TTV_TRY(ttvk_sq_err(recon + c0, target + c0, sizes + c0, n, dtype, clamp, acc, (hipStream_t)stream));
  }
  return TTV_OK;
}

int ttv_debug_set(int flags) {
  g_ttv_debug = flags;
  return TTV_OK;
}

int ttv_debug_stamps(void* device_buffer) {
  g_ttv_stamps = (long long*)device_buffer;
  return TTV_OK;
}

int ttv_prof_begin(int kernel_class, int max_records) {
  TTV_CHECK_ARG(g_ttv_prof_class == 0, "prof_begin: already recording");
  TTV_CHECK_ARG(kernel_class > 0 && max_records > 0 && max_records <= (1 << 20), "prof_begin: bad arguments");
  g_prof_start = new hipEvent_t[max_records];
  g_prof_stop = new hipEvent_t[max_records];
  for (int i = 0; i < max_records; ++i) {
    if (hipEventCreate(&g_prof_start[i]) != hipSuccess || hipEventCreate(&g_prof_stop[i]) != hipSuccess) {
      ttv_set_error("prof_begin: hipEventCreate failed");
      return TTV_ERR_LAUNCH;
    }
  }
  g_prof_cap = max_records;
  g_prof_n = 0;
  g_ttv_prof_class = kernel_class;
  return TTV_OK;
}

int ttv_prof_end(double* total_ms, int* count) {
  TTV_CHECK_ARG(g_ttv_prof_class != 0 && total_ms && count, "prof_end: not recording");
  g_ttv_prof_class = 0;
  double tot = 0.0;
  for (int i = 0; i < g_prof_n; ++i) {
    float ms = 0.f;
    (void)hipEventSynchronize(g_prof_stop[i]);
    (void)hipEventElapsedTime(&ms, g_prof_start[i], g_prof_stop[i]);
    tot += ms;
  }
  *total_ms = tot;
  *count = g_prof_n;
  for (int i = 0; i < g_prof_cap; ++i) {
    (void)hipEventDestroy(g_prof_start[i]);
    (void)hipEventDestroy(g_prof_stop[i]);
  }
  delete[] g_prof_start;
  delete[] g_prof_stop;
  g_prof_start = g_prof_stop = nullptr;
  g_prof_cap = g_prof_n = 0;
  return TTV_OK;
}

int ttv_codebook_histogram(const int32_t* indices, int n, int64_t* counts, int codebook_size, void* stream) {
  TTV_CHECK_ARG(n == 0 || (indices && counts), "codebook_histogram: null buffer");
  return ttvk_histogram(indices, n, counts, codebook_size, (hipStream_t)stream);
}

}  // extern "C"

// Training step of a tower (reference train.py:65-83 = autograd through blocks.py / transformer.py / fsq.py):
// a tape-recording forward (unfused kernel sequence, every tensor backward needs is written into the caller's tape)
// and the hand-written backward launch sequence.  Gradients of parameters are ACCUMULATED into caller-zeroed fp32
// buffers (atomics in the reduction kernels); activation gradients travel in the compute dtype, the residual-stream
// gradient in fp32.
#include <stdio.h>
#include <stdlib.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

#define TTV_TRY(expr)                \
  do {                               \
    int rc__ = (expr);               \
    if (rc__ != TTV_OK) return rc__; \
  } while (0)

static inline int64_t al(int64_t v) { return (v + 255) & ~(int64_t)255; }
static inline int esz(int dt) { return dt == TTV_BF16 ? 2 : 4; }

struct Tape {
  char* base;
  int64_t total;
  // per tower
  char *patches, *pe, *hpre, *pn;     // encoder: gathered patches, proj_in output (pre-norm); decoder: pre-norm latent rows, ln_post out
  char* X[65];                        // residual stream at every layer boundary (X[0] = tower input rows, X[layers] = output)
  struct L { char *xn1, *qkvg, *a, *ag, *x1, *xn2, *u, *h, *y1, *y2; float* lse; } l[64];   // y1 / y2: the KEEL sums, dtype tape_y_dtype()
  char *agc, *xc;                     // encoder, last layer: its latent rows of ag and of X[layers - 1], compact (see latent_tail)
};

// The KEEL sums y = alpha * x + f(x) the post-norms read (forward) and differentiate (backward) are kept in the tower's dtype: a bf16
// tower rounds them to bf16 as the reference's autocast does (transformer.py:141: bf16 * alpha + bf16) and as the inference path does -
// half the bytes on the GEMM that writes them, the norm that reads them and the backward chain (round 5; fp32 before).  The opt-in
// fused-norm tape forward (TTV_TRAIN_FUSED_NORMS=1) writes fp32 sums from its GEMM kernel and keeps fp32.
static bool tape_fuse_norms() {
  static const bool v = getenv("TTV_TRAIN_FUSED_NORMS") && getenv("TTV_TRAIN_FUSED_NORMS")[0] == '1';
  return v;
}
static int tape_y_dtype(int dt) {
  static const bool f32 = getenv("TTV_TAPE_Y_F32") && getenv("TTV_TAPE_Y_F32")[0] == '1';      // A/B: fp32 sums as before round 5
  return (dt == TTV_F32 || tape_fuse_norms() || f32) ? TTV_F32 : TTV_BF16;
}
static Tape carve_tape(const ttv_tower_dims* d, const ttv_batch* b, char* base) {
  Tape t;
  const int64_t ye = esz(tape_y_dtype(d->dtype));
  const int64_t e = esz(d->dtype), L = b->total_rows, P = b->sum_patches, K = b->sum_tokens, dm = d->width;
  const int64_t g = (int64_t)d->kv_heads * d->head_dim, nq = 2 * dm + 2 * g;
  const int64_t pd = (int64_t)d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return p; };
  t.base = base;
  t.patches = t.pe = t.hpre = t.pn = nullptr;
  t.agc = t.xc = nullptr;
  if (d->kind == TTV_ENCODER) { t.patches = take(P * pd * e); t.pe = take(P * dm * e); t.agc = take(K * dm * e); t.xc = take(K * dm * e); }
  else { t.hpre = take(K * dm * e); t.pn = take(P * dm * e); }
  for (int i = 0; i <= d->layers; ++i) t.X[i] = take(L * dm * e);
  for (int i = 0; i < d->layers; ++i) {
    Tape::L& l = t.l[i];
    l.xn1 = take(L * dm * e); l.qkvg = take(L * nq * e); l.lse = (float*)take(L * d->q_heads * 4);
    l.a = take(L * dm * e); l.ag = take(L * dm * e);
    l.y1 = take(i > 0 ? L * dm * ye : 0); l.x1 = take(L * dm * e); l.xn2 = take(L * dm * e);
    l.u = take(L * 2 * d->inner * e); l.h = take(L * d->inner * e); l.y2 = take(i > 0 ? L * dm * ye : 0);
  }
  t.total = off;
  return t;
}

struct BwdWs {
  float *dxa, *dxb, *delta, *dkv, *colsum, *small_f32;
  float* dxc;       // [sum_tokens, width] fp32: the latent rows of the residual-stream gradient, compact (encoder, last layer)
  char *g_d, *g_d2, *g_i, *g_2i, *g_nq, *g_pd;
  char *g_df2, *g_do;   // [L, width]: the second df buffer (layers alternate) and do - operands the weight-gradient stream may still read
  float* wg_part;
  int64_t wg_part_bytes;
  int64_t total;
};
static BwdWs carve_bwd(const ttv_tower_dims* d, const ttv_batch* b, char* base) {
  BwdWs w;
  const int64_t e = esz(d->dtype), L = b->total_rows, P = b->sum_patches, dm = d->width;
  const int64_t g = (int64_t)d->kv_heads * d->head_dim, nq = 2 * dm + 2 * g;
  const int64_t pd = (int64_t)d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return p; };
  w.dxa = (float*)take(L * dm * 4); w.dxb = (float*)take(L * dm * 4);
  w.delta = (float*)take(L * d->q_heads * 4);
  w.dkv = (float*)take(d->dtype == TTV_F32 ? L * 2 * g * 4 : 0);
  w.colsum = (float*)take(dm * 4);
  w.small_f32 = (float*)take(L * dm * 4);          // fp32 [rows, d] scratch for the d <-> token_size projections
  w.dxc = (float*)take((int64_t)b->sum_tokens * dm * 4);
  w.g_d = take(L * dm * e); w.g_d2 = take(L * dm * e); w.g_i = take(L * d->inner * e); w.g_2i = take(L * 2 * d->inner * e);
  w.g_nq = take(L * nq * e); w.g_pd = take(P * pd * e);
  w.g_df2 = take(L * dm * e); w.g_do = take(L * dm * e);
  // split partial tiles of the weight-gradient GEMMs (largest of the shapes the tower uses)
  int64_t wgb = 0;
  if (d->dtype == TTV_BF16) {
    // the four weight gradients of a layer keep their partial tiles side by side and are summed by one launch (WgradBatch): room for
    // their SUM; the head / tail projections run on their own (largest single shape)
    const int64_t shapes[6][3] = {{L, dm, d->inner}, {L, 2 * d->inner, dm}, {L, dm, dm}, {L, nq, dm}, {P, dm, pd}, {P, pd, dm}};
    int64_t layer_sum = 0;
    for (int i = 0; i < 6; ++i) {
      const int64_t v = ttvk_wgrad_ws_bytes((int)shapes[i][0], (int)shapes[i][1], (int)shapes[i][2]);
      if (i < 4) layer_sum += v;
      wgb = v > wgb ? v : wgb;
    }
    wgb = layer_sum > wgb ? layer_sum : wgb;
  }
  w.wg_part = (float*)take(wgb); w.wg_part_bytes = wgb;
  w.total = off;
  return w;
}

static int check(const ttv_tower_dims* d, const ttv_batch* b) {
  TTV_CHECK_ARG(d && b, "null dims/batch");
  TTV_CHECK_ARG(d->dtype == TTV_BF16 || d->dtype == TTV_F32, "bad dtype");
  TTV_CHECK_ARG(d->layers >= 1 && d->layers <= 64, "layers out of range");
  TTV_CHECK_ARG(d->head_dim == 64 && d->width == d->q_heads * 64 && d->width <= 1024, "width must be q_heads*64 <= 1024");
  TTV_CHECK_ARG(b->blocks64 && b->row_seq, "training needs batch.blocks64 and batch.row_seq");
  TTV_CHECK_ARG(d->token_size >= 1 && d->token_size <= TTV_MAX_TOKEN, "training towers take token_size <= %d", TTV_MAX_TOKEN);
  return TTV_OK;
}

// The encoder's output is read from its latent rows only (blocks.py:101-103): in its LAST layer everything behind the attention - out_proj,
// the KEEL norms, the feed-forward - runs on the sum K_b latent rows, gathered into compact buffers; the tape entries of that sub-layer
// (x1, xn2, u, h, y1, y2) then hold those rows in their first sum K_b rows, and the backward gathers / scatters accordingly.  The gradient
// of the loss with respect to the last layer's patch-row outputs is zero, so nothing is lost; attention (forward and backward) still runs
// on every row - a query row with dO = 0 contributes nothing.  TTV_ENC_LATENT_LAST=0 / ttv_debug_set bit 19: every row (A/B, tests).
static bool latent_tail(const ttv_tower_dims* d, const ttv_batch* b, int layer) {
  static const bool env = !(getenv("TTV_ENC_LATENT_LAST") && getenv("TTV_ENC_LATENT_LAST")[0] == '0');
  return env && !(g_ttv_debug & 524288) && d->kind == TTV_ENCODER && layer == d->layers - 1 && b->latent_rows && b->sum_tokens > 0 &&
         b->sum_tokens < b->total_rows && d->inner >= d->width;
}

// ... and the attention itself for the latent QUERY rows only (keys / values: every row), forward and backward, when the batch carries the
// latent work table and the clip descriptors (the backward derives K_b from them)
static bool latent_attn(const ttv_tower_dims* d, const ttv_batch* b, int layer) {
  return latent_tail(d, b, layer) && b->qblocks_latent && b->n_qblocks_latent > 0 && b->clip_desc;
}

// ------------------------------------------------------------------------------------------------ forward (tape)
static int layers_forward_train(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_batch* b, Tape& t, hipStream_t s) {
  const int L = b->total_rows, dm = d->width, g = d->kv_heads * d->head_dim, dt = d->dtype, nq = 2 * dm + 2 * g, I = d->inner;
  // Round 4, OPT-IN (TTV_TRAIN_FUSED_NORMS=1): where the fused residual + KEEL-norm GEMM applies (bf16, width 256) the post-norm, the fp32
  // pre-norm sum the backward needs AND the following pre-norm come out of the GEMM kernel (GemmArgs.sum_f32 / y2): 5 launches per layer
  // instead of 9, same tape.  Measured neutral (same box, tools/bench_train.py: 8.234 vs 8.236 ms at 32 clips, 3.47 vs 3.44 ms at 5): the
  // row-owning GEMM kernels lose against the tiled GEMM + 160-token tiles what the four row kernels cost, and the step is GPU-bound, not
  // launch-bound - so the proven sequence stays the default.
  const bool fuse_norms = tape_fuse_norms();
  const int ydt = tape_y_dtype(dt);
  bool xn1_ready = false;          // this layer's xn1 was written by the layer below
  for (int i = 0; i < d->layers; ++i) {
    const ttv_layer_weights& lw = w->layers[i];
    Tape::L& l = t.l[i];
    if (!xn1_ready) TTV_TRY(ttvk_rmsnorm(t.X[i], dt, dm, nullptr, l.xn1, dt, dm, nullptr, lw.pre_ln, L, dm, d->eps, s));
    xn1_ready = false;
    GemmArgs a = {};
    a.dtype = dt; a.x = l.xn1; a.ldx = dm; a.w = lw.to_qkv; a.ldw = dm; a.M = L; a.N = nq; a.K = dm; a.y = l.qkvg; a.ldy = nq;
    a.rope_cs = b->rope_cs; a.rope_q_end = dm; a.rope_k_begin = 2 * dm; a.rope_k_end = 2 * dm + g;
    a.rope_ids = b->rope_ids; a.rope_base = b->rope_ids ? b->rope_base : nullptr;
    TTV_TRY(ttvk_gemm(EPI_QKV_ROPE, a, s));
    // encoder, last layer: attention outputs are needed for the latent query rows only (latent_attn); the raw output of the other rows is
    // zeroed (the gate backward reads it for every row, times a zero gradient) and the backward skips those query rows too
    const bool lat_q = latent_attn(d, b, i);
    const int32_t* qb = lat_q ? b->qblocks_latent : b->qblocks;
    const int nqb = lat_q ? b->n_qblocks_latent : b->n_qblocks;
    const int pair = (!lat_q && b->qblocks_paired) ? TTV_ATTN_PAIRED : 0;
    if (lat_q) (void)hipMemsetAsync(l.a, 0, (size_t)L * dm * esz(dt), s);
    if (dt == TTV_BF16) {
      // one launch writes the raw output a (tape) and the gated one ag = a * sigmoid(gate) (gate applied to the stored, rounded a)
      TTV_TRY(ttvk_attention(l.qkvg, nq, l.ag, dm, b->cu_seqlens, qb, nqb, d->q_heads, d->kv_heads, d->head_dim, TTV_ATTN_GATE | pair, dt, s, l.lse, l.a));
    } else {
      TTV_TRY(ttvk_attention(l.qkvg, nq, l.a, dm, b->cu_seqlens, qb, nqb, d->q_heads, d->kv_heads, d->head_dim, pair, dt, s, l.lse));
      TTV_TRY(ttvk_gate_fwd(l.a, dm, (const char*)l.qkvg + (size_t)dm * esz(dt), nq, l.ag, dm, L, dm, dt, s));
    }
    // the rest of the layer on (Lc rows: ag_in, x_in): every row, or - encoder, last layer - the latent rows, compact
    const bool lat = latent_tail(d, b, i);
    const int Lc = lat ? b->sum_tokens : L;
    char* ag_in = l.ag;
    char* x_in = t.X[i];
    char* x_out = t.X[i + 1];
    if (lat) {
      TTV_TRY(ttvk_copy_rows(l.ag, (int64_t)dm * esz(dt), b->latent_rows, t.agc, (int64_t)dm * esz(dt), nullptr, Lc, dm * esz(dt), s));
      TTV_TRY(ttvk_copy_rows(t.X[i], (int64_t)dm * esz(dt), b->latent_rows, t.xc, (int64_t)dm * esz(dt), nullptr, Lc, dm * esz(dt), s));
      ag_in = t.agc; x_in = t.xc; x_out = t.xc;          // the layer's output: compact in place of its input rows, scattered below
    }
    GemmArgs o = {};
    o.dtype = dt; o.x = ag_in; o.ldx = dm; o.w = lw.out_proj; o.ldw = dm; o.M = Lc; o.N = dm; o.K = dm; o.resid = x_in; o.ldr = dm;
    if (i == 0) {
      o.alpha = 1.f; o.y = l.x1; o.ldy = dm;
      TTV_TRY(ttvk_gemm(EPI_RESID_T, o, s));
    } else if (fuse_norms && ttvk_gemm_supports_resid_norm(dt, dm, dm)) {
      o.alpha = d->alpha; o.y = l.x1; o.ldy = dm; o.norm_gain = lw.attn_post_ln; o.eps = d->eps;
      o.sum_f32 = (float*)l.y1; o.ld_sum = dm; o.y2 = l.xn2; o.ldy2 = dm; o.norm_gain2 = lw.ffd_norm;
      TTV_TRY(ttvk_gemm(EPI_RESID_NORM, o, s));
    } else {
      o.alpha = d->alpha; o.y = l.y1; o.ldy = dm;
      TTV_TRY(ttvk_gemm(ydt == TTV_F32 ? EPI_RESID_F32 : EPI_RESID_T, o, s));
      TTV_TRY(ttvk_rmsnorm(l.y1, ydt, dm, nullptr, l.x1, dt, dm, nullptr, lw.attn_post_ln, Lc, dm, d->eps, s));
    }
    if (!(i > 0 && fuse_norms && ttvk_gemm_supports_resid_norm(dt, dm, dm)))
      TTV_TRY(ttvk_rmsnorm(l.x1, dt, dm, nullptr, l.xn2, dt, dm, nullptr, lw.ffd_norm, Lc, dm, d->eps, s));
    GemmArgs f = {};
    if (dt == TTV_BF16) {
      // one launch: u = xn2 W12^T kept for the backward (through `resid`) and h = gelu(gate) * x from the stored values
      f.dtype = dt; f.x = l.xn2; f.ldx = dm; f.w = lw.w12; f.ldw = dm; f.M = Lc; f.N = I; f.K = dm; f.y = l.h; f.ldy = I;
      f.resid = l.u; f.ldr = 2 * I;
      TTV_TRY(ttvk_gemm(EPI_GEGLU, f, s));
    } else {
      f.dtype = dt; f.x = l.xn2; f.ldx = dm; f.w = lw.w12; f.ldw = dm; f.M = Lc; f.N = 2 * I; f.K = dm; f.y = l.u; f.ldy = 2 * I;
      TTV_TRY(ttvk_gemm(EPI_STORE, f, s));
      TTV_TRY(ttvk_geglu_fwd(l.u, 2 * I, l.h, I, Lc, I, dt, s));
    }
    GemmArgs f3 = {};
    f3.dtype = dt; f3.x = l.h; f3.ldx = I; f3.w = lw.w3; f3.ldw = I; f3.M = Lc; f3.N = dm; f3.K = I; f3.resid = l.x1; f3.ldr = dm;
    if (i == 0) {
      f3.alpha = 1.f; f3.y = x_out; f3.ldy = dm;
      TTV_TRY(ttvk_gemm(EPI_RESID_T, f3, s));
    } else if (fuse_norms && ttvk_gemm_supports_resid_norm(dt, dm, I)) {
      f3.alpha = d->alpha; f3.y = x_out; f3.ldy = dm; f3.norm_gain = lw.ffd_post_ln; f3.eps = d->eps;
      f3.sum_f32 = (float*)l.y2; f3.ld_sum = dm;
      if (i + 1 < d->layers) {          // the next layer's pre-norm rides along
        f3.y2 = t.l[i + 1].xn1; f3.ldy2 = dm; f3.norm_gain2 = w->layers[i + 1].pre_ln;
        xn1_ready = true;
      }
      TTV_TRY(ttvk_gemm(EPI_RESID_NORM, f3, s));
    } else {
      f3.alpha = d->alpha; f3.y = l.y2; f3.ldy = dm;
      TTV_TRY(ttvk_gemm(ydt == TTV_F32 ? EPI_RESID_F32 : EPI_RESID_T, f3, s));
      TTV_TRY(ttvk_rmsnorm(l.y2, ydt, dm, nullptr, x_out, dt, dm, nullptr, lw.ffd_post_ln, Lc, dm, d->eps, s));
    }
    if (lat)     // the latent rows of the tower's output where the tail (and the backward) read them; its patch rows are never read
      TTV_TRY(ttvk_copy_rows(t.xc, (int64_t)dm * esz(dt), nullptr, t.X[i + 1], (int64_t)dm * esz(dt), b->latent_rows, Lc, dm * esz(dt), s));
  }
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------ weight-gradient stream
// The weight-gradient GEMMs of a layer (dW = dY^T X, four per layer, plus the launch that sums their split partial tiles) are off the
// backward's critical path: nothing of the layer reads them.  They run on a second HIP stream, forked from the caller's stream by an event
// behind the producer of each dY and joined at the end of the layer, so that their launch tails and partly filled rounds overlap the
// dX chain (and, at the reference's 5-clip batches, their launch latencies).  The caller's stream stays the only one the caller sees:
// every fork is an event recorded on it, the join makes it wait - legal inside a HIP-graph capture of the step as well.
// Operands the second stream reads are not overwritten before the join (df alternates between two buffers, do / da have their own).
// TTV_WGRAD_SIDE=0: everything on the caller's stream (A/B).
struct WgradSide {
  hipStream_t w = nullptr;
  hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t join = nullptr;
  int state = 0;   // 0 not tried, 1 ready, -1 unavailable
};
static WgradSide* wgrad_side(bool dp) {
  // 1 (default): on, also with the DP path's per-layer events attached; 0: never; 3: single-process steps only.  (For a while the DP
  // path kept the one-stream backward after a red run of the two-rank test; that run turned out to be the test's own tolerance - an
  // AdamW update of a near-zero gradient element is rounding noise - and showed up again with this stream off: DESIGN 5b.)
  static const int mode = getenv("TTV_WGRAD_SIDE") ? atoi(getenv("TTV_WGRAD_SIDE")) : 1;
  if (mode <= 0 || (dp && mode == 3)) return nullptr;
  static thread_local WgradSide tab[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  WgradSide& sd = tab[dev];
  if (sd.state == 0) {
    sd.state = -1;
    if (hipStreamCreateWithFlags(&sd.w, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&sd.fork[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    sd.state = 1;
  }
  return sd.state == 1 ? &sd : nullptr;
}
// the second stream takes up the caller's stream at this point
static int side_fork(WgradSide* sd, int k, hipStream_t s) {
  if (!sd) return TTV_OK;
  if (hipEventRecord(sd->fork[k], s) != hipSuccess || hipStreamWaitEvent(sd->w, sd->fork[k], 0) != hipSuccess) {
    ttv_set_error("backward: fork of the weight-gradient stream failed");
    return TTV_ERR_LAUNCH;
  }
  return TTV_OK;
}
static int side_join(WgradSide* sd, hipStream_t s) {
  if (!sd) return TTV_OK;
  if (hipEventRecord(sd->join, sd->w) != hipSuccess || hipStreamWaitEvent(s, sd->join, 0) != hipSuccess) {
    ttv_set_error("backward: join of the weight-gradient stream failed");
    return TTV_ERR_LAUNCH;
  }
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------ backward (layers)
// On entry ws.dxa holds dL/dX[layers] (fp32); on exit ws.dxa holds dL/dX[0].
static int layers_backward(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_tower_weights_t* wt, const ttv_batch* b, Tape& t,
                           const ttv_tower_grads* gr, BwdWs& ws, hipStream_t s) {
  const int L = b->total_rows, dm = d->width, g = d->kv_heads * d->head_dim, dt = d->dtype, nq = 2 * dm + 2 * g, I = d->inner;
  const int ydt = tape_y_dtype(dt);
  float* dx = ws.dxa;     // gradient of the layer output
  float* tmp = ws.dxb;
  WgradBatch wgb;
  char* g_df[2] = {ws.g_d, ws.g_df2};   // df of this layer / da of this layer, then df of the layer below
  int cur = 0;
  for (int i = d->layers - 1; i >= 0; --i) {
    const ttv_layer_weights& lw = w->layers[i];
    const ttv_layer_weights_t& lt = wt->layers[i];
    const ttv_layer_grads& lg = gr->layers[i];
    Tape::L& l = t.l[i];
    // the layer's weight-gradient GEMMs and their summing launch: on the second stream (wgrad_side) when any is wanted
    WgradSide* sd = (lg.w3 || lg.w12 || lg.out_proj || lg.to_qkv) ? wgrad_side(gr->layer_done_events != nullptr) : nullptr;
    const hipStream_t sw = sd ? sd->w : s;
    char* const df = g_df[cur];
    char* const da = g_df[cur ^ 1];
    // ---------------- feed-forward sub-layer: X[i+1] = post_ln(alpha*x1 + w3 h)  (layer 0: x1 + w3 h) ----------------
    // The top layer undoes its feed-forward post-norm here; for the layers below it was chained onto the pre_ln backward of the
    // layer above (end of the previous iteration).
    // encoder, last layer: the sub-layers behind the attention ran on the latent rows only (latent_tail): their backward does too, on the
    // gathered latent rows of dL/dX[layers] (every other row of it is zero)
    const bool lat = latent_tail(d, b, i);
    const int Lc = lat ? b->sum_tokens : L;
    const long nLc = (long)Lc * dm;
    float* const dxf = dx;                 // the full-size gradient buffer
    if (lat) {
      TTV_TRY(ttvk_copy_rows(dxf, (int64_t)dm * 4, b->latent_rows, ws.dxc, (int64_t)dm * 4, nullptr, Lc, dm * 4, s));
      dx = ws.dxc;
    }
    float* dx1;
    if (i == d->layers - 1) {
      if (i > 0) {
        TTV_TRY(ttvk_rmsnorm_bwd(l.y2, ydt, dm, nullptr, dx, TTV_F32, dm, nullptr, lw.ffd_post_ln, tmp, TTV_F32, dm, nullptr, 0, lg.ffd_post_ln, Lc, dm, d->eps, s));
        // tmp = dy2 ; df (T) = dy2 ; dx1 = alpha * dy2 (into dx)
        TTV_TRY(ttvk_scale_cast(tmp, d->alpha, dx, df, dt, nLc, s));
      } else {
        TTV_TRY(ttvk_scale_cast(dx, 1.f, nullptr, df, dt, nLc, s));   // df = (T) dx ; dx1 = dx
      }
    }
    dx1 = dx;
    // dh = df W3 ; dW3 += df^T h
    GemmArgs a = {};
    a.dtype = dt; a.x = df; a.ldx = dm; a.w = lt.w3_t; a.ldw = dm; a.M = Lc; a.N = I; a.K = dm; a.y = ws.g_i; a.ldy = I;
    TTV_TRY(side_fork(sd, 0, s));          // df is complete
    TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
    TTV_TRY(ttvk_wgrad(df, dm, l.h, I, lg.w3, I, Lc, dm, I, dt, ws.wg_part, ws.wg_part_bytes, sw, &wgb));
    TTV_TRY(ttvk_geglu_bwd(l.u, 2 * I, ws.g_i, I, ws.g_2i, 2 * I, Lc, I, dt, s));
    TTV_TRY(side_fork(sd, 1, s));          // du is complete
    // dxn2 = du W12 ; dW12 += du^T xn2
    GemmArgs c = {};
    c.dtype = dt; c.x = ws.g_2i; c.ldx = 2 * I; c.w = lt.w12_t; c.ldw = 2 * I; c.M = Lc; c.N = dm; c.K = 2 * I; c.y = ws.g_d2; c.ldy = dm;
    TTV_TRY(ttvk_gemm(EPI_STORE, c, s));
    TTV_TRY(ttvk_wgrad(ws.g_2i, 2 * I, l.xn2, dm, lg.w12, dm, Lc, 2 * I, dm, dt, ws.wg_part, ws.wg_part_bytes, sw, &wgb));
    // dx1 += rmsnorm_bwd(x1, ffd_norm, dxn2), then straight through the attention sub-layer's post-norm:
    // ---------------- attention sub-layer: x1 = post_ln(alpha*x + out_proj ag) ----------------
    // i > 0: dy1 = rmsnorm_bwd(y1, attn_post_ln, dx1) ; do = (T) dy1 ; dx = alpha * dy1      i == 0: do = (T) dx1 ; dx = dx1
    TTV_TRY(ttvk_rmsnorm_bwd_chain(l.x1, dm, ws.g_d2, dm, lw.ffd_norm, lg.ffd_norm, dx1, dm, i > 0 ? l.y1 : nullptr, ydt, dm,
                                   i > 0 ? lw.attn_post_ln : nullptr, i > 0 ? lg.attn_post_ln : nullptr, i > 0 ? d->alpha : 1.f, ws.g_do, dm, Lc,
                                   dm, d->eps, dt, s));
    TTV_TRY(side_fork(sd, 2, s));          // do is complete
    // dag = do Wo ; dWo += do^T ag
    GemmArgs e = {};
    e.dtype = dt; e.x = ws.g_do; e.ldx = dm; e.w = lt.out_proj_t; e.ldw = dm; e.M = Lc; e.N = dm; e.K = dm; e.y = ws.g_d2; e.ldy = dm;
    TTV_TRY(ttvk_gemm(EPI_STORE, e, s));
    TTV_TRY(ttvk_wgrad(ws.g_do, dm, lat ? t.agc : l.ag, dm, lg.out_proj, dm, Lc, dm, dm, dt, ws.wg_part, ws.wg_part_bytes, sw, &wgb));
    // back to every row: the gradient of the attention output and of the residual stream are zero outside the latent rows
    char* dag = ws.g_d2;
    if (lat) {
      const size_t es_ = esz(dt);
      (void)hipMemsetAsync(ws.g_i, 0, (size_t)L * dm * es_, s);
      TTV_TRY(ttvk_copy_rows(ws.g_d2, (int64_t)dm * es_, nullptr, ws.g_i, (int64_t)dm * es_, b->latent_rows, Lc, dm * (int)es_, s));
      dag = ws.g_i;
      (void)hipMemsetAsync(dxf, 0, (size_t)L * dm * sizeof(float), s);
      TTV_TRY(ttvk_copy_rows(ws.dxc, (int64_t)dm * 4, nullptr, dxf, (int64_t)dm * 4, b->latent_rows, Lc, dm * 4, s));
      dx = dxf;
    }
    // da = dag*sigmoid(gate) (into the df buffer this layer does not use) ; dgate -> dqkvg[:, d:2d]
    char* dqkvg = ws.g_nq;
    const size_t es = esz(dt);
    TTV_TRY(ttvk_gate_bwd(dag, dm, l.a, dm, (const char*)l.qkvg + (size_t)dm * es, nq, da, dm, dqkvg + (size_t)dm * es, nq, L, dm, dt,
                          ws.delta, s));   // also fills delta = sum_d da * a per (row, head) for the attention backward
    // attention backward -> dq, dk, dv columns of dqkvg
    TTV_TRY(ttvk_attention_bwd(l.qkvg, nq, l.a, dm, da, dm, l.lse, ws.delta, b->cu_seqlens, b->blocks64, b->n_blocks64, b->row_seq, dqkvg, nq,
                               ws.dkv, L, d->q_heads, d->kv_heads, dt, b->rope_cs, s, 1, latent_attn(d, b, i) ? b->clip_desc : nullptr));   // dq, dk come back un-rotated; delta from the gate backward
    // dxn1 = dqkvg Wqkv ; dWqkv += dqkvg^T xn1
    GemmArgs q = {};
    q.dtype = dt; q.x = dqkvg; q.ldx = nq; q.w = lt.to_qkv_t; q.ldw = nq; q.M = L; q.N = dm; q.K = nq; q.y = ws.g_d2; q.ldy = dm;
    TTV_TRY(side_fork(sd, 3, s));          // dqkvg is complete
    TTV_TRY(ttvk_gemm(EPI_STORE, q, s));
    TTV_TRY(ttvk_wgrad(dqkvg, nq, l.xn1, dm, lg.to_qkv, dm, L, nq, dm, dt, ws.wg_part, ws.wg_part_bytes, sw, &wgb));
    TTV_TRY(ttvk_wgrad_flush(&wgb, sw));     // the layer's four weight gradients: one summing launch
    // dx += rmsnorm_bwd(X[i], pre_ln, dxn1) = dL/dX[i]; chained with the head of the layer below: through its feed-forward
    // post-norm (layers >= 1) and the bf16 copy df that its w3 products read
    if (i == 0) {
      TTV_TRY(ttvk_rmsnorm_bwd(t.X[i], dt, dm, nullptr, ws.g_d2, dt, dm, nullptr, lw.pre_ln, dx, TTV_F32, dm, nullptr, 1, lg.pre_ln, L, dm, d->eps, s));
    } else {
      const bool post = i - 1 > 0;
      TTV_TRY(ttvk_rmsnorm_bwd_chain(t.X[i], dm, ws.g_d2, dm, lw.pre_ln, lg.pre_ln, dx, dm, post ? t.l[i - 1].y2 : nullptr, ydt, dm,
                                     post ? w->layers[i - 1].ffd_post_ln : nullptr, post ? gr->layers[i - 1].ffd_post_ln : nullptr,
                                     post ? d->alpha : 1.f, g_df[cur ^ 1], dm, L, dm, d->eps, dt, s));   // df of the layer below (da is consumed)
    }
    cur ^= 1;
    TTV_TRY(side_join(sd, s));             // the caller's stream takes the weight gradients back in; dY operands may be overwritten from here
    // every gradient of layer i is final here (its ffd_post_ln gain received its contribution in the iteration above)
    if (gr->layer_done_events && gr->layer_done_events[i]) {
      if (hipEventRecord((hipEvent_t)gr->layer_done_events[i], s) != hipSuccess) {
        ttv_set_error("backward: hipEventRecord(layer_done_events[%d]) failed", i);
        return TTV_ERR_LAUNCH;
      }
    }
  }
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------ optimizer step
// Gradient-norm clip + AdamW over a list of parameter tensors in two launches (reference train.py:76-77 clip_gradients, :183-190 AdamW;
// what torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW compute).  torch's multi-tensor path takes 7 launches and ~230 us per
// step for the 7 M parameters of the tiny tokenizer (98 MB of traffic in bf16: 0.4 TB/s); these two move the same bytes once:
//   k_opt_gradsq: one block per 8 192-element chunk of a tensor -> partial[chunk] = sum g^2 (fixed order inside the block)
//   k_opt_adamw : every block first sums ALL partials in one fixed order (the same value in every block, bit-reproducible run to run -
//                 no atomics), derives the clip factor min(1, max_norm / (norm + 1e-6)), then updates its chunk; the gradient is scaled in
//                 registers and NOT rewritten (clip_grad_norm_ scales p.grad in place: the only difference a caller could see).
// fp32 arithmetic whatever the tensors' dtype (fp32 or bf16; state in the parameter's dtype), operation order of torch's fused kernel:
//   p -= lr wd p ; m = lerp(m, g, 1 - b1) ; v = b2 v + (1 - b2) g g ; p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps).
#define OPT_CHUNK 8192
struct OptEntry { void* p; const void* g; void* m; void* v; long long n; };
template <typename T> struct OptVec;
template <> struct OptVec<float> {
  static __device__ __forceinline__ void load(const float* q, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
  static __device__ __forceinline__ void store(float* q, const float (&o)[8]) {
    *reinterpret_cast<f32x4*>(q) = (f32x4){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(q + 4) = (f32x4){o[4], o[5], o[6], o[7]};
  }
};
template <> struct OptVec<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* q, float (&o)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(q);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
  }
  static __device__ __forceinline__ void store(bf16_t* q, const float (&o)[8]) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (bf16_t)o[e];
    *reinterpret_cast<bf16x8*>(q) = a;
  }
};
// block-wide sum in a fixed order: lanes by DPP / permlane inside the wave, the four waves through LDS
__device__ __forceinline__ float opt_block_sum(float v, float* red) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
template <typename T>
__global__ __launch_bounds__(256) void k_opt_gradsq(const OptEntry* __restrict__ tab, const int2* __restrict__ chunks, float* __restrict__ partial) {
  __shared__ float red[4];
  const int2 c = chunks[blockIdx.x];
  const OptEntry e = tab[c.x];
  const T* g = (const T*)e.g;
  const long long end = e.n < (long long)c.y + OPT_CHUNK ? e.n : (long long)c.y + OPT_CHUNK;
  float acc = 0.f;
  if (((uintptr_t)g & 15) == 0) {
    long long i = (long long)c.y + threadIdx.x * 8;
    for (; i + 8 <= end; i += 256 * 8) {
      float v[8];
      OptVec<T>::load(g + i, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fmaf(v[k], v[k], acc);
    }
    if (i < end)
      for (long long j = i; j < end; ++j) { const float v = (float)g[j]; acc = fmaf(v, v, acc); }
  } else {
    for (long long j = (long long)c.y + threadIdx.x; j < end; j += 256) { const float v = (float)g[j]; acc = fmaf(v, v, acc); }
  }
  const float t = opt_block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
struct OptHyper { float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, max_norm; };
__device__ __forceinline__ void opt_update(float& p, float g, float& m, float& v, const OptHyper& h, float step_size) {
  p -= h.lr * h.wd * p;
  const float w = 1.0f - h.beta1;
  m = w < 0.5f ? m + w * (g - m) : g - (g - m) * (1.0f - w);        // at::native::lerp
  v = h.beta2 * v + (1.0f - h.beta2) * g * g;
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p -= step_size * m / denom;
}
template <typename T>
__global__ __launch_bounds__(256) void k_opt_adamw(const OptEntry* __restrict__ tab, const int2* __restrict__ chunks, const float* __restrict__ partial,
                                                   int n_partial, OptHyper h, float* __restrict__ out_norm) {
  __shared__ float red[4];
  float coef = 1.0f;
  if (n_partial > 0) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
    const float norm = sqrtf(opt_block_sum(acc, red));
    if (out_norm && blockIdx.x == 0 && threadIdx.x == 0) out_norm[0] = norm;
    if (h.max_norm > 0.f) coef = fminf(1.0f, h.max_norm / (norm + 1e-6f));
  }
  const int2 c = chunks[blockIdx.x];
  const OptEntry e = tab[c.x];
  T* p = (T*)e.p; const T* g = (const T*)e.g; T* m = (T*)e.m; T* v = (T*)e.v;
  const long long end = e.n < (long long)c.y + OPT_CHUNK ? e.n : (long long)c.y + OPT_CHUNK;
  const float step_size = h.lr / h.bc1;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
    long long i = (long long)c.y + threadIdx.x * 8;
    for (; i + 8 <= end; i += 256 * 8) {
      float pv[8], gv[8], mv[8], vv[8];
      OptVec<T>::load(p + i, pv); OptVec<T>::load(g + i, gv); OptVec<T>::load(m + i, mv); OptVec<T>::load(v + i, vv);
#pragma unroll
      for (int k = 0; k < 8; ++k) opt_update(pv[k], gv[k] * coef, mv[k], vv[k], h, step_size);
      OptVec<T>::store(p + i, pv); OptVec<T>::store(m + i, mv); OptVec<T>::store(v + i, vv);
    }
    if (i < end)
      for (long long j = i; j < end; ++j) {
        float pv = (float)p[j], mv = (float)m[j], vv = (float)v[j];
        opt_update(pv, (float)g[j] * coef, mv, vv, h, step_size);
        p[j] = (T)pv; m[j] = (T)mv; v[j] = (T)vv;
      }
  } else {
    for (long long j = (long long)c.y + threadIdx.x; j < end; j += 256) {
      float pv = (float)p[j], mv = (float)m[j], vv = (float)v[j];
      opt_update(pv, (float)g[j] * coef, mv, vv, h, step_size);
      p[j] = (T)pv; m[j] = (T)mv; v[j] = (T)vv;
    }
  }
}

extern "C" {

int64_t ttv_tower_tape_bytes(const ttv_tower_dims* dims, const ttv_batch* batch) {
  if (!dims || !batch) return -1;
  return carve_tape(dims, batch, nullptr).total;
}
int64_t ttv_tower_bwd_workspace_bytes(const ttv_tower_dims* dims, const ttv_batch* batch) {
  if (!dims || !batch) return -1;
  return carve_bwd(dims, batch, nullptr).total;
}

int ttv_fsq_backward(const ttv_fsq_params* p, const float* z, const void* dcodes, int dcodes_dtype, float* dz, int rows, void* stream) {
  TTV_CHECK_ARG(p && (rows == 0 || (z && dcodes && dz)), "fsq_backward: null buffer");
  return ttvk_fsq_bwd(p, z, dcodes, dcodes_dtype, dz, rows, (hipStream_t)stream);
}

int ttv_encoder_forward_train(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_batch* b, const void* const* clips, float* z,
                              void* tape, int64_t tape_bytes, void* stream) {
  TTV_TRY(check(d, b));
  TTV_CHECK_ARG(d->kind == TTV_ENCODER && w && w->layers && clips && z && tape, "encoder_forward_train: bad argument");
  hipStream_t s = (hipStream_t)stream;
  Tape t = carve_tape(d, b, (char*)tape);
  TTV_CHECK_ARG(t.total <= tape_bytes, "encoder_forward_train: tape too small");
  const int dm = d->width, dt = d->dtype, P = b->sum_patches;
  const int pd = d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  for (int c0 = 0; c0 < b->n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
    const int n = b->n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? b->n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
    TTV_TRY(ttvk_patch_copy(false, (void* const*)(clips + c0), b->clip_desc, c0, n, d->patch_t, d->patch_h, d->patch_w, d->pix_channels, t.patches, pd, dt, b->max_patches_per_clip, s));
  }
  GemmArgs a = {};
  a.dtype = dt; a.x = t.patches; a.ldx = pd; a.w = w->proj_in_w; a.ldw = pd; a.M = P; a.N = dm; a.K = pd; a.y = t.pe; a.ldy = dm;
  a.bias = w->proj_in_b; a.add_scalar = w->mask_token;
  TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
  TTV_TRY(ttvk_rmsnorm(t.pe, dt, dm, nullptr, t.X[0], dt, dm, b->patch_rows, w->ln_pre_p, P, dm, d->eps, s));
  TTV_TRY(ttvk_fill_const_rows(t.X[0], dt, dm, b->latent_rows, b->sum_tokens, dm, w->mask_token, w->ln_pre_t, d->eps, s));
  TTV_TRY(layers_forward_train(d, w, b, t, s));
  TTV_TRY(ttvk_enc_tail(t.X[d->layers], dt, dm, b->latent_rows, b->sum_tokens, dm, w->ln_post, d->eps, w->proj_out_w, w->proj_out_b, d->token_size, nullptr, z, nullptr, nullptr, nullptr, s));
  return TTV_OK;
}

int ttv_encoder_backward(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_tower_weights_t* wt, const ttv_batch* b, const float* dz,
                         void* tape, const ttv_tower_grads* gr, void* const* dclips, void* workspace, int64_t workspace_bytes, void* stream) {
  TTV_TRY(check(d, b));
  TTV_CHECK_ARG(d->kind == TTV_ENCODER && w && wt && wt->layers && (!gr || gr->layers) && dz && tape && workspace, "encoder_backward: bad argument");
  // grads == NULL: every parameter is frozen (generator step through the discriminator, loss_module.py:144-151) - only the
  // input-clip gradient is produced; all weight-gradient GEMMs, gain / bias reductions are skipped
  static const ttv_layer_grads no_layer_grads[64] = {};
  static const ttv_tower_grads no_grads = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, no_layer_grads};
  if (!gr) {
    TTV_CHECK_ARG(dclips, "encoder_backward: nothing to compute (no parameter and no input gradients requested)");
    gr = &no_grads;
  }
  hipStream_t s = (hipStream_t)stream;
  Tape t = carve_tape(d, b, (char*)tape);
  BwdWs ws = carve_bwd(d, b, (char*)workspace);
  TTV_CHECK_ARG(ws.total <= workspace_bytes, "encoder_backward: workspace too small");
  const int L = b->total_rows, dm = d->width, dt = d->dtype, P = b->sum_patches, K = b->sum_tokens, C = d->token_size;
  const int pd = d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  // ---- tail: z = proj_out(n) + b, n = ln_post(X_last[latent rows]) ----
  TTV_TRY(ttvk_rmsnorm(t.X[d->layers], dt, dm, b->latent_rows, ws.g_d, dt, dm, nullptr, w->ln_post, K, dm, d->eps, s));   // n
  TTV_TRY(ttvk_outer_small(dz, TTV_F32, C, C, ws.g_d, dt, dm, nullptr, gr->proj_out_w, dm, 0, K, dm, s));
  TTV_TRY(ttvk_colsum(dz, TTV_F32, C, nullptr, K, C, gr->proj_out_b, s));
  TTV_TRY(ttvk_expand_small(dz, TTV_F32, C, C, w->proj_out_w, dt, dm, 1, ws.g_d2, dt, dm, K, dm, s));                      // dn
  (void)hipMemsetAsync(ws.dxa, 0, (size_t)L * dm * sizeof(float), s);
  TTV_TRY(ttvk_rmsnorm_bwd(t.X[d->layers], dt, dm, b->latent_rows, ws.g_d2, dt, dm, nullptr, w->ln_post, ws.dxa, TTV_F32, dm, b->latent_rows, 0, gr->ln_post, K, dm, d->eps, s));
  TTV_TRY(layers_backward(d, w, wt, b, t, gr, ws, s));
  // ---- head ----
  // latent rows are the constant vector ln_pre_t(mask_token * 1)
  (void)hipMemsetAsync(ws.colsum, 0, dm * sizeof(float), s);
  TTV_TRY(ttvk_colsum(ws.dxa, TTV_F32, dm, b->latent_rows, K, dm, ws.colsum, s));
  TTV_TRY(ttvk_const_rows_bwd(ws.colsum, w->mask_token, w->ln_pre_t, dt, d->eps, gr->ln_pre_t, gr->mask_token, dm, s));
  // patch rows: X0[patch] = ln_pre_p(pe), pe = proj_in(patches) + bias + mask_token
  TTV_TRY(ttvk_rmsnorm_bwd(t.pe, dt, dm, nullptr, ws.dxa, TTV_F32, dm, b->patch_rows, w->ln_pre_p, ws.g_d, dt, dm, nullptr, 0, gr->ln_pre_p, P, dm, d->eps, s));  // dpe
  TTV_TRY(ttvk_sumall(ws.g_d, dt, dm, nullptr, P, dm, nullptr, 1.f, gr->mask_token, s));
  TTV_TRY(ttvk_colsum(ws.g_d, dt, dm, nullptr, P, dm, gr->proj_in_b, s));
  TTV_TRY(ttvk_wgrad(ws.g_d, dm, t.patches, pd, gr->proj_in_w, pd, P, dm, pd, dt, ws.wg_part, ws.wg_part_bytes, s));
  if (dclips) {
    GemmArgs a = {};
    a.dtype = dt; a.x = ws.g_d; a.ldx = dm; a.w = wt->proj_in_t; a.ldw = dm; a.M = P; a.N = pd; a.K = dm; a.y = ws.g_pd; a.ldy = pd;
    TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
    for (int c0 = 0; c0 < b->n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
      const int n = b->n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? b->n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
      TTV_TRY(ttvk_patch_copy(true, dclips + c0, b->clip_desc, c0, n, d->patch_t, d->patch_h, d->patch_w, d->pix_channels, ws.g_pd, pd, dt, b->max_patches_per_clip, s));
    }
  }
  return TTV_OK;
}

int ttv_decoder_forward_train(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_batch* b, const void* codes, void* const* clips_out,
                              void* tape, int64_t tape_bytes, void* workspace, int64_t workspace_bytes, void* stream) {
  TTV_TRY(check(d, b));
  TTV_CHECK_ARG(d->kind == TTV_DECODER && w && w->layers && codes && clips_out && tape && workspace, "decoder_forward_train: bad argument");
  hipStream_t s = (hipStream_t)stream;
  Tape t = carve_tape(d, b, (char*)tape);
  TTV_CHECK_ARG(t.total <= tape_bytes, "decoder_forward_train: tape too small");
  const int dm = d->width, dt = d->dtype, P = b->sum_patches;
  const int pd = d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  TTV_CHECK_ARG((int64_t)P * pd * esz(dt) <= workspace_bytes, "decoder_forward_train: workspace too small");
  TTV_TRY(ttvk_dec_embed_ex(codes, d->token_size, w->proj_in_w, w->proj_in_b, w->mask_token, w->ln_pre_t, t.X[0], dt, dm, b->latent_rows, b->sum_tokens, dm, d->eps, t.hpre, s));
  TTV_TRY(ttvk_fill_const_rows(t.X[0], dt, dm, b->patch_rows, P, dm, w->mask_token, w->ln_pre_p, d->eps, s));
  TTV_TRY(layers_forward_train(d, w, b, t, s));
  TTV_TRY(ttvk_rmsnorm(t.X[d->layers], dt, dm, b->patch_rows, t.pn, dt, dm, nullptr, w->ln_post, P, dm, d->eps, s));
  GemmArgs a = {};
  a.dtype = dt; a.x = t.pn; a.ldx = dm; a.w = w->proj_out_w; a.ldw = dm; a.M = P; a.N = pd; a.K = dm; a.y = workspace; a.ldy = pd;
  a.bias = w->proj_out_b;
  TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
  for (int c0 = 0; c0 < b->n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
    const int n = b->n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? b->n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
    TTV_TRY(ttvk_patch_copy(true, clips_out + c0, b->clip_desc, c0, n, d->patch_t, d->patch_h, d->patch_w, d->pix_channels, workspace, pd, dt, b->max_patches_per_clip, s));
  }
  return TTV_OK;
}

int ttv_decoder_backward(const ttv_tower_dims* d, const ttv_tower_weights* w, const ttv_tower_weights_t* wt, const ttv_batch* b, const void* codes,
                         const void* const* dclips_out, void* tape, const ttv_tower_grads* gr, float* dcodes, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  TTV_TRY(check(d, b));
  TTV_CHECK_ARG(d->kind == TTV_DECODER && w && wt && wt->layers && gr && gr->layers && codes && dclips_out && tape && workspace, "decoder_backward: bad argument");
  hipStream_t s = (hipStream_t)stream;
  Tape t = carve_tape(d, b, (char*)tape);
  BwdWs ws = carve_bwd(d, b, (char*)workspace);
  TTV_CHECK_ARG(ws.total <= workspace_bytes, "decoder_backward: workspace too small");
  const int L = b->total_rows, dm = d->width, dt = d->dtype, P = b->sum_patches, K = b->sum_tokens, C = d->token_size;
  const int pd = d->pix_channels * d->patch_t * d->patch_h * d->patch_w;
  // ---- tail: clips = unpatch(proj_out(pn) + bias), pn = ln_post(X_last[patch rows]) ----
  for (int c0 = 0; c0 < b->n_clips; c0 += TTV_MAX_CLIPS_PER_LAUNCH) {
    const int n = b->n_clips - c0 < TTV_MAX_CLIPS_PER_LAUNCH ? b->n_clips - c0 : TTV_MAX_CLIPS_PER_LAUNCH;
    TTV_TRY(ttvk_patch_copy(false, (void* const*)(dclips_out + c0), b->clip_desc, c0, n, d->patch_t, d->patch_h, d->patch_w, d->pix_channels, ws.g_pd, pd, dt, b->max_patches_per_clip, s));
  }
  TTV_TRY(ttvk_colsum(ws.g_pd, dt, pd, nullptr, P, pd, gr->proj_out_b, s));
  TTV_TRY(ttvk_wgrad(ws.g_pd, pd, t.pn, dm, gr->proj_out_w, dm, P, pd, dm, dt, ws.wg_part, ws.wg_part_bytes, s));
  GemmArgs a = {};
  a.dtype = dt; a.x = ws.g_pd; a.ldx = pd; a.w = wt->proj_out_t; a.ldw = pd; a.M = P; a.N = dm; a.K = pd; a.y = ws.g_d2; a.ldy = dm;   // dpn
  TTV_TRY(ttvk_gemm(EPI_STORE, a, s));
  (void)hipMemsetAsync(ws.dxa, 0, (size_t)L * dm * sizeof(float), s);
  TTV_TRY(ttvk_rmsnorm_bwd(t.X[d->layers], dt, dm, b->patch_rows, ws.g_d2, dt, dm, nullptr, w->ln_post, ws.dxa, TTV_F32, dm, b->patch_rows, 0, gr->ln_post, P, dm, d->eps, s));
  TTV_TRY(layers_backward(d, w, wt, b, t, gr, ws, s));
  // ---- head ----
  (void)hipMemsetAsync(ws.colsum, 0, dm * sizeof(float), s);
  TTV_TRY(ttvk_colsum(ws.dxa, TTV_F32, dm, b->patch_rows, P, dm, ws.colsum, s));
  TTV_TRY(ttvk_const_rows_bwd(ws.colsum, w->mask_token, w->ln_pre_p, dt, d->eps, gr->ln_pre_p, gr->mask_token, dm, s));
  // latent rows: X0[latent] = ln_pre_t(hpre), hpre = proj_in(codes) + bias + mask_token
  TTV_TRY(ttvk_rmsnorm_bwd(t.hpre, dt, dm, nullptr, ws.dxa, TTV_F32, dm, b->latent_rows, w->ln_pre_t, ws.small_f32, TTV_F32, dm, nullptr, 0, gr->ln_pre_t, K, dm, d->eps, s));  // dh
  TTV_TRY(ttvk_sumall(ws.small_f32, TTV_F32, dm, nullptr, K, dm, nullptr, 1.f, gr->mask_token, s));
  TTV_TRY(ttvk_colsum(ws.small_f32, TTV_F32, dm, nullptr, K, dm, gr->proj_in_b, s));
  TTV_TRY(ttvk_outer_small(codes, dt, C, C, ws.small_f32, TTV_F32, dm, nullptr, gr->proj_in_w, C, 1, K, dm, s));
  if (dcodes) TTV_TRY(ttvk_reduce_small(ws.small_f32, TTV_F32, dm, nullptr, w->proj_in_w, dt, C, C, dcodes, C, K, dm, s));
  return TTV_OK;
}

int64_t ttv_linear_wgrad_workspace_bytes(int L, int N, int K) { return ttvk_wgrad_ws_bytes(L, N, K); }

int ttv_linear_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int L, int N, int K, int dtype, void* workspace,
                     int64_t workspace_bytes, void* stream) {
  TTV_CHECK_ARG(L == 0 || (dy && x && dw), "linear_wgrad: null buffer");
  TTV_CHECK_ARG(workspace_bytes >= 0 && (workspace || workspace_bytes == 0), "linear_wgrad: bad workspace");
  return ttvk_wgrad(dy, lddy, x, ldx, dw, lddw, L, N, K, dtype, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

int ttv_rmsnorm_backward_chain(const void* x, int ldx, const void* dy, int lddy, const float* gain1, float* dgain1, float* dx, int lddx,
                               const float* y, int ldy, const float* gain2, float* dgain2, float out_scale, void* cast_out, int ldc, int rows,
                               int width, float eps, int dtype, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (x && dy && gain1 && dx), "rmsnorm_backward_chain: null buffer");
  return ttvk_rmsnorm_bwd_chain(x, ldx, dy, lddy, gain1, dgain1, dx, lddx, y, TTV_F32, ldy, gain2, dgain2, out_scale, cast_out, ldc, rows, width, eps,
                                dtype, (hipStream_t)stream);
}

int ttv_rmsnorm_backward(const void* x, int ldx, const void* dy, int lddy, const float* gain, void* dx, int lddx, float* dgain, int rows,
                         int width, float eps, int dtype, void* stream) {
  TTV_CHECK_ARG(rows == 0 || (x && dy && gain && dx), "rmsnorm_backward: null buffer");
  return ttvk_rmsnorm_bwd(x, dtype, ldx, nullptr, dy, dtype, lddy, nullptr, gain, dx, dtype, lddx, nullptr, 0, dgain, rows, width, eps, (hipStream_t)stream);
}

int ttv_attention_backward(const void* qkvg, int ld, const void* o, int ldo, const void* dout, int ldd, const float* lse, float* delta,
                           const int32_t* cu_seqlens, const int32_t* blocks64, int n_blocks64, const int32_t* row_seq, void* dqkvg,
                           int ldg, float* dkv_scratch, int total_rows, int q_heads, int kv_heads, int dtype, const float* rope_cs,
                           void* stream) {
  TTV_CHECK_ARG(total_rows == 0 || (qkvg && o && dout && lse && delta && cu_seqlens && blocks64 && dqkvg), "attention_backward: null buffer");
  return ttvk_attention_bwd(qkvg, ld, o, ldo, dout, ldd, lse, delta, cu_seqlens, blocks64, n_blocks64, row_seq, dqkvg, ldg, dkv_scratch,
                            total_rows, q_heads, kv_heads, dtype, rope_cs, (hipStream_t)stream);
}

int ttv_attention_lse(const void* qkvg, int ld, void* out, int ldo, const int32_t* cu_seqlens, const int32_t* qblocks, int n_qblocks,
                      int q_heads, int kv_heads, int head_dim, int flags, int dtype, float* lse, void* stream) {
  TTV_CHECK_ARG(n_qblocks == 0 || (qkvg && out && cu_seqlens && qblocks), "attention_lse: null buffer");
  return ttvk_attention(qkvg, ld, out, ldo, cu_seqlens, qblocks, n_qblocks, q_heads, kv_heads, head_dim, flags, dtype, (hipStream_t)stream, lse);
}

int ttv_opt_grad_sumsq(const void* table, const int32_t* chunks, int n_chunks, int dtype, float* partials, void* stream) {
  if (n_chunks == 0) return TTV_OK;
  TTV_CHECK_ARG(n_chunks > 0 && table && chunks && partials, "opt_grad_sumsq: null buffer");
  TTV_CHECK_ARG(dtype == TTV_BF16 || dtype == TTV_F32, "opt_grad_sumsq: dtype");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_opt_gradsq<bf16_t>), dim3(n_chunks), dim3(256), 0, s, (const OptEntry*)table, (const int2*)chunks, partials);
  else hipLaunchKernelGGL((k_opt_gradsq<float>), dim3(n_chunks), dim3(256), 0, s, (const OptEntry*)table, (const int2*)chunks, partials);
  TTV_CHECK_LAUNCH("opt_grad_sumsq");
  return TTV_OK;
}

int ttv_opt_adamw_step(const void* table, const int32_t* chunks, int n_chunks, int dtype, const float* partials, int n_partials, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2_sqrt,
                       float max_norm, float* out_norm, void* stream) {
  if (n_chunks == 0) return TTV_OK;
  TTV_CHECK_ARG(n_chunks > 0 && table && chunks, "opt_adamw_step: null buffer");
  TTV_CHECK_ARG(n_partials == 0 || partials, "opt_adamw_step: partials missing");
  TTV_CHECK_ARG(dtype == TTV_BF16 || dtype == TTV_F32, "opt_adamw_step: dtype");
  TTV_CHECK_ARG(bias_correction1 > 0.f && bias_correction2_sqrt > 0.f, "opt_adamw_step: bias corrections must be positive");
  hipStream_t s = (hipStream_t)stream;
  const OptHyper h = {lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, max_norm};
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_opt_adamw<bf16_t>), dim3(n_chunks), dim3(256), 0, s, (const OptEntry*)table, (const int2*)chunks, partials, n_partials, h, out_norm);
  else hipLaunchKernelGGL((k_opt_adamw<float>), dim3(n_chunks), dim3(256), 0, s, (const OptEntry*)table, (const int2*)chunks, partials, n_partials, h, out_norm);
  TTV_CHECK_LAUNCH("opt_adamw_step");
  return TTV_OK;
}

}  // extern "C"

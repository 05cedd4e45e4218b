/* titok_hip.h - C-ABI of libtitok_hip.so: the MI355X (gfx950) compute path of the TiTok-Video tokenizer.
 *
 * The reference (NilanEkanayake/TiTok-Video) exposes this path as a Python nn.Module API, not an FFI
 * (SURVEY.md section 8b).  The Python mirror in titok_video_amd/model/ keeps that API and binds these
 * entry points with ctypes; every entry point below names the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, floats, an opaque hipStream_t passed as void*.  No torch types.
 *   - every call only ENQUEUES work on `stream` and never synchronises; no hidden global state, no allocation:
 *     the caller owns all buffers including the workspace (size from ttv_tower_workspace_bytes).
 *   - return value: 0 = TTV_OK, otherwise a TTV_ERR_* code; ttv_error_string() explains the last error of
 *     the calling thread.  The Python shim raises RuntimeError on any non-zero code.
 *   - activations/weights of linear layers are in the compute dtype (TTV_BF16 or TTV_F32); RMSNorm gains,
 *     mask_token, rope tables and all statistics are fp32; matrix products accumulate in fp32.
 *   - rows of a packed batch: for clip b, rows [cu[b], cu[b]+K_b) are its latent tokens, rows
 *     [cu[b]+K_b, cu[b+1]) its patch tokens in (t,h,w) raster order (model/base/blocks.py:85-86).
 */
#ifndef TITOK_HIP_H
#define TITOK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTV_OK 0
#define TTV_ERR_INVALID 1      /* bad argument / unsupported shape */
#define TTV_ERR_LAUNCH 2       /* hip launch error */
#define TTV_ERR_UNSUPPORTED 3

#define TTV_BF16 0
#define TTV_F32 1

#define TTV_ENCODER 0
#define TTV_DECODER 1

#define TTV_MAX_FSQ 8
#define TTV_MAX_TOKEN 64       /* widest latent token (encoder proj_out rows / decoder proj_in columns): FSQ needs <= TTV_MAX_FSQ, the
                                  nearest-codebook-entry quantiser of BASELINE configs #4 / #5 uses 32 / 64 */
#define TTV_MAX_CLIPS_PER_LAUNCH 64

const char* ttv_error_string(void);
int ttv_version(void);

/* ---- FSQ (model/quantizer/fsq.py) -------------------------------------------------------------------- */
typedef struct ttv_fsq_params {
  int32_t n;                        /* codebook_dim = len(levels)                         fsq.py:69 */
  int32_t levels[TTV_MAX_FSQ];      /* _levels                                            fsq.py:63 */
  int32_t basis[TTV_MAX_FSQ];       /* _basis = cumprod([1]+levels[:-1])                  fsq.py:66 */
  float half_l[TTV_MAX_FSQ];        /* (levels-1)*(1+eps)/2, fp32, computed by the host   fsq.py:80 */
  float offset[TTV_MAX_FSQ];        /* 0.5 for even levels                                fsq.py:81 */
  float shift[TTV_MAX_FSQ];         /* atanh(offset/half_l)                               fsq.py:82 */
  float half_width[TTV_MAX_FSQ];    /* levels // 2                                        fsq.py:89 */
} ttv_fsq_params;

/* FSQ.forward (fsq.py:123-135): z [rows,n] (dtype) -> codes [rows,n] (dtype), indices int32 [rows];
 * bounded (fp32 [rows,n], value before rounding) is optional (NULL to skip). */
int ttv_fsq_forward(const ttv_fsq_params* p, const void* z, int z_dtype, int rows, void* codes, int codes_dtype,
                    int32_t* indices, float* bounded, void* stream);
/* FSQ.indices_to_codes (fsq.py:100-121): int32 [rows] -> codes [rows,n] (dtype). */
int ttv_fsq_indices_to_codes(const ttv_fsq_params* p, const int32_t* indices, int rows, void* codes, int codes_dtype,
                             void* stream);

/* ---- nearest-codebook-entry (L2) quantiser ---------------------------------------------------------------
 * Not a reference component (the reference quantises with FSQ only); BASELINE.json's north_star / configs #4, #5 ask for it.  On
 * the FSQ lattice implicit_codebook * (levels // 2) (fsq.py:73-76) applied to FSQ.bound(z) (fsq.py:78-83) it returns FSQ's indices
 * (fsq.py:105-109) away from rounding ties; for learned / synthetic codebooks it is argmin_n ||z - c_n||^2 with the LOWEST index on
 * exact ties (torch.argmin's rule).  z [rows, C], codebook [N, C] in `dtype` (fp32: exact-fp32 MFMA; bf16: bf16 MFMA, fp32 sums),
 * C <= 64.  cnorm: fp32 [N] = ||c_n||^2 from ttv_vq_codebook_norms (once per codebook).  best_dist (optional) fp32 [rows]. */
int ttv_vq_codebook_norms(const void* codebook, int dtype, int ld, int N, int C, float* cnorm, void* stream);
/* workspace: ttv_vq_workspace_bytes(rows) bytes, 8-byte aligned (per-row merge keys of the codebook splits; cleared by the call). */
int64_t ttv_vq_workspace_bytes(int rows);
int ttv_vq_l2_argmin(const void* z, int dtype, int ldz, const void* codebook, int ldc, const float* cnorm, int rows, int N, int C,
                     int32_t* indices, float* best_dist, void* workspace, int64_t workspace_bytes, void* stream);
/* straight-through lookup: codes[r] = codebook[indices[r]] (the value the decoder sees; gradients pass to z unchanged). */
int ttv_vq_lookup(const void* codebook, int dtype, int ldc, const int32_t* indices, int rows, int C, void* codes, int ldo, void* stream);
/* Backward of the lookup with respect to the codebook: dcodebook[indices[r], :] += dcodes[r, :] (fp32 accumulation, float atomics; the
 * caller zeroes dcodebook [N, C], ldc).  With codes = codebook[idx] + (z - stopgrad(z)) this is the codebook's gradient of the
 * straight-through quantiser named by BASELINE.json's north_star; the encoder's is the identity (FSQ's round_ste, fsq.py:48-51, is the
 * reference's instance of the same estimator). */
int ttv_vq_lookup_backward(const void* dcodes, int dtype, int ld, const int32_t* indices, int rows, int C, float* dcodebook, int ldc, void* stream);

/* ---- mixed bf16 / fp8 linears (BASELINE config #5; not a reference feature: the reference runs bf16 autocast) -------------
 * Row-wise OCP e4m3 quantisation: y = gain ? RMSNorm(x) * gain (eps) : x;  scales[r] = max|y_r| / 448;  out[r] = e4m3(y_r / scales[r]).
 * in [rows, width] (dtype, leading dim ld_in), out uint8 [rows, ld_out], width % 4 == 0, width <= 1024. */
int ttv_quant_rows_fp8(const void* in, int dtype, int ld_in, const float* gain, float eps, void* out, int ld_out, float* scales, int rows,
                       int width, void* stream);
/* y[M,N] (bf16) = (xq * x_scale[:,None]) @ (wq * w_scale[:,None])^T on v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulation), K % 128 == 0.
 * epilogue: 0 plain store; 1 to_qkv + rotary (rope_cs, d_model, gqa_dim as ttv_linear_qkv_rope; N = 2 d_model + 2 gqa_dim);
 * 2 GEGLU (wq [2N, K], y[:, f] = gelu(acc[:, N+f]) * acc[:, f], as ttv_linear_geglu). */
int ttv_linear_fp8(const void* xq, int ldx, const float* x_scale, const void* wq, int ldw, const float* w_scale, void* y, int ldy, int M, int N,
                   int K, int epilogue, const float* rope_cs, int d_model, int gqa_dim, void* stream);

/* Split image of an fp32 matrix [rows, K] (K % 4 == 0) for the three-pass bf16 linears (ttv_tower_weights.f32_split3): for every aligned
 * group of four k values the 16 bytes (hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3) with hi = bf16(x), lo = bf16(x - hi) (round to nearest even).
 * out has the size and leading dimension (in fp32 elements) of the input.  The image is what k_gemm_f32<.., SPLIT> stages with the copies
 * of the fp32 kernel; no reference counterpart (the reference computes in the parameter dtype, titok.py:61). */
int ttv_split3_pack(const float* w, int ldw, void* out, int ldo, int rows, int K, void* stream);
/* y[M,N] fp32 = x[M,K] fp32 @ W^T (+ bias fp32) with W given as its split image: the three-pass bf16 linear on its own (tests; the towers
 * reach the same kernel with their own epilogues).  Replaces nn.Linear (blocks.py:49,93; transformer.py:73-76) in the split-bf16 mode. */
int ttv_linear_split3(const float* x, int ldx, const void* w_image, int ldw, const float* bias, float* y, int ldy, int M, int N, int K, void* stream);

/* Block-scaled (OCP MX) e4m3 quantisation of rows [rows, width] (dtype bf16 / fp32, width % 128 == 0): q[r, k] = round_e4m3(y / 2^E),
 * one E8M0 byte E + 127 per 32 consecutive k, E = ceil(log2(max|block| / 448)); with row_scales != NULL (weights) y = in / row_scales[r],
 * row_scales[r] = max|row| / 448, else y = in.  mx: ttv_mx_scale_bytes_per_row(width) bytes per row, block b = k / 32 at byte
 * (b & 3) * nkp + (b >> 2), nkp = round_up(width / 128, 4) (the order the MFMA lanes of ttv_linear_fp8_mx read them in).
 * Not a reference feature (the reference runs bf16 autocast, configs/tiny.yaml:70): BASELINE.json configs[4] "mixed bf16/fp8 MFMA". */
int64_t ttv_mx_scale_bytes_per_row(int width);
int ttv_quant_mx_fp8(const void* in, int dtype, int ld_in, void* out, int ld_out, void* mx, float* row_scales, int rows, int width, void* stream);
/* y[M,N] (bf16) = epilogue( (xq * 2^Ex) (wq * 2^Ew)^T * x_row_scale[m] * w_row_scale[n] ) on v_mfma_scale_f32_16x16x128_f8f6f4 with the
 * block scales as the instruction's scale operands; x_row_scale / w_row_scale may be NULL (= 1).  epilogue 0 store, 1 qkv + rotary
 * (as ttv_linear_fp8), 2 GEGLU (wq [2N, K]), 3 y = alpha * resid + acc (resid [M,N] bf16, ldr; y may alias resid).
 * Replaces the linears of transformer.py:47-56,86-104 in the mixed-precision configuration. */
int ttv_linear_fp8_mx(const void* xq, int ldx, const void* x_mx, const float* x_row_scale, const void* wq, int ldw, const void* w_mx,
                      const float* w_row_scale, void* y, int ldy, int M, int N, int K, int epilogue, const float* rope_cs, int d_model,
                      int gqa_dim, const void* resid, int ldr, float alpha, void* stream);

/* ---- single ops (exported for parity tests; the tower entry points below chain them) ------------------ */

/* RMSNorm (flash_attn RMSNorm as used at blocks.py:51-52,66 / transformer.py:42,77,122-123):
 * out[dst_rows[i]] = in[src_rows[i]] * rsqrt(mean(in^2)+eps) * gain, fp32 math.  Row maps may be NULL (identity). */
int ttv_rmsnorm(const void* in, int in_dtype, int ld_in, const int32_t* src_rows, void* out, int out_dtype, int ld_out,
                const int32_t* dst_rows, const float* gain, int rows, int width, float eps, void* stream);

/* apply_rotary_emb (model/base/rope.py:19-27) on x [rows, heads, 64] in place; rope_cs fp32 [rows,64] =
 * (cos[32] | sin[32]) per row, entries 30,31 = (1,0) so the last 4 dims of a head stay untouched. */
int ttv_rope_apply(void* x, int dtype, int ld, int rows, int heads, const float* rope_cs, void* stream);

/* y[M,N] = x[M,K] @ w[N,K]^T (+ bias[N]) (+ *add_scalar): nn.Linear (blocks.py:93,166,173; transformer.py:49,55,87,104). */
int ttv_linear(const void* x, int ldx, const void* w, int ldw, const void* bias, const float* add_scalar, void* y, int ldy,
               int M, int N, int K, int dtype, void* stream);

/* to_qkv + rotary (transformer.py:87,97-98): y[M, 2d+2g] = x @ w^T with apply_rotary_emb fused on the q columns
 * [0,d) and k columns [2d,2d+g); rope_cs as in ttv_rope_apply. */
int ttv_linear_qkv_rope(const void* x, int ldx, const void* w, int ldw, void* y, int ldy, int M, int d_model, int gqa_dim,
                        const float* rope_cs, int dtype, void* stream);
/* GEGLU w12 + activation (transformer.py:49-52): w [2I,K]; y[M,I] = gelu_erf(x@w[I:]^T) * (x@w[:I]^T). */
int ttv_linear_geglu(const void* x, int ldx, const void* w, int ldw, void* y, int ldy, int M, int I, int K, int dtype,
                     void* stream);
/* out_proj / w3 + residual (transformer.py:129-130,141,144): y = alpha*resid + x@w^T; y is fp32 when y_f32 != 0
 * (the KEEL pre-norm sum), else dtype (in-place on resid allowed). */
int ttv_linear_residual(const void* x, int ldx, const void* w, int ldw, const void* resid, int ldr, float alpha, void* y,
                        int ldy, int y_f32, int M, int N, int K, int dtype, void* stream);

/* The whole KEEL step of one sub-layer (transformer.py:141-142 / 144-145) in one kernel:
 * y = RMSNorm(alpha*resid + x@w^T) * gain, stored in dtype (y may alias resid).  Returns TTV_ERR_UNSUPPORTED unless a
 * full-row kernel exists for the shape (bf16, N == 256, K % 8 == 0); callers then use ttv_linear_residual + ttv_rmsnorm. */
int ttv_linear_residual_norm(const void* x, int ldx, const void* w, int ldw, const void* resid, int ldr, float alpha,
                             const float* gain, float eps, void* y, int ldy, int M, int N, int K, int dtype, void* stream);

/* Whole GEGLU sub-layer (transformer.py:47-56) + residual/KEEL step (:130, :144-145) in one kernel, bf16, width 256:
 * y = [RMSNorm](alpha*x + (gelu(xn@w12[I:]^T) * (xn@w12[:I]^T)) @ w3^T) [* post_gain], xn = RMSNorm(x)*norm_gain.
 * mlp_packed = ttv_mlp_pack(w12 * norm_gain[None,:], w3, out_proj, next_qkv_folded, next_qkv_rows): the panel images the
 * kernels stream by LDS-DMA (built once per weight version, ttv_mlp_pack_bytes(inner, next_qkv_rows) bytes; out_proj and
 * next_qkv_folded may be NULL / 0 when the parts that use them are not); post_gain NULL = plain residual (layer 0).
 * y may alias x.  TTV_ERR_UNSUPPORTED for other dtypes/widths. */
int64_t ttv_mlp_pack_bytes(int inner, int next_qkv_rows);
int ttv_mlp_pack(const void* w12_folded, const void* w3, const void* out_proj, const void* next_qkv_folded, int next_qkv_rows, int inner,
                 int width, int dtype, void* mlp_packed, void* stream);
int ttv_mlp_fused(const void* x, int ldx, const void* mlp_packed, int inner, void* y, int ldy, const float* post_gain, float alpha,
                  float eps, int M, int width, int dtype, void* stream);
/* Optional last part of ttv_layer_tail_fused: the NEXT layer's attention input (transformer.py:86-98),
 * qkv = rotary(RMSNorm(y) @ (to_qkv * pre_ln_gain)^T), rows = 2d+2g of the folded weight packed by ttv_mlp_pack. */
typedef struct ttv_next_qkv {
  void* qkv; int32_t ld;        /* [M, ld] output (q | gate | k | v) */
  const float* rope_cs;         /* [M, 64] (cos | sin) */
  int32_t rows;                 /* % 64 == 0 */
  int32_t rope_q_end, rope_k_begin, rope_k_end;   /* rotary applies to features [0, q_end) and [k_begin, k_end); whole heads */
} ttv_next_qkv;
/* Everything of a transformer layer after the attention kernel (transformer.py:104, 129-130 / 141-145) in one kernel:
 *   x1 = [RMSNorm](attn_alpha*x + ao@out_proj^T) [* attn_post_gain]          (gain NULL = plain residual, layer 0)
 *   y  = [RMSNorm](ffd_alpha*x1 + GEGLU-feed-forward(x1)) [* ffd_post_gain]   (as ttv_mlp_fused)
 *   next->qkv = the next layer's rotated qkv projection of y                  (next NULL = not computed)
 * y may alias x (it is also used to hand x1 from one wave to its partner inside a workgroup).  Same support as ttv_mlp_fused. */
int ttv_layer_tail_fused(const void* ao, int ldao, const float* attn_post_gain, float attn_alpha, const void* x, int ldx,
                         const void* mlp_packed, int inner, void* y, int ldy, const float* ffd_post_gain, float ffd_alpha, float eps,
                         int M, int width, int dtype, const ttv_next_qkv* next, void* stream);

/* Token / patch initialisation of the towers (blocks.py:95-97 encoder, :165-167 decoder), exported for parity tests:
 * constant rows x[rows_map[i]] = RMSNorm(mask_token * 1_d) * gain - the encoder's latent rows (ln_pre_t) and the decoder's patch
 * rows (ln_pre_p); */
int ttv_fill_const_rows(void* x, int dtype, int ld, const int32_t* rows_map, int rows, int width, const float* mask_token,
                        const float* gain, float eps, void* stream);
/* the decoder's latent rows x[rows_map[i]] = RMSNorm(proj_in(codes[i]) + mask_token) * gain (blocks.py:125,165-166):
 * codes [rows, token_size] (dtype), w [width, token_size], bias [width] (dtype). */
int ttv_decoder_embed(const void* codes, int token_size, const void* w, const void* bias, const float* mask_token, const float* gain,
                      void* x, int dtype, int ld, const int32_t* rows_map, int rows, int width, float eps, void* stream);

/* flash_attn_varlen_func as called at transformer.py:100, fused with the sigmoid gate of transformer.py:103:
 * qkvg [L, 2d+2g] packed (q | gate | k | v) with RoPE already applied to q,k; out [L,d] = attn * sigmoid(gate).
 * Non-causal, block-diagonal over cu_seqlens (device int32 [n_seq+1]), GQA, softmax scale head_dim^-0.5.
 * qblocks: device int32 [n_qblocks,4] work table = (sequence id, first query row within the sequence, q-head, mode), one
 * entry per query block per q-head; sequence id -1 = padding.  mode 0: 128 query rows; mode 1 ("half item"): 64 query rows
 * with the key range split between the two wave pairs of the block and merged at the end - about half the duration of a
 * full item, used by the host to fill the tail of the grid at a finer grain.  The host orders the table so that
 * entries i, i+8, i+16, ... (one XCD under round-robin dispatch) share a (sequence, kv-head): its K/V are then fetched into
 * one L2 only; half items come last.
 * flags: bit 0 (TTV_ATTN_GATE) multiply by sigmoid(gate), else the raw attention output is written; bit 1 (TTV_ATTN_PAIRED,
 * bf16 only) the table is PAIRED: with the flat table read as rows of 8 list slots (entry i belongs to list i % 8), entries
 * 2j and 2j+1 of a list describe the same (sequence, query rows, mode) for two q-heads of one kv-head; one 8-wave block then
 * computes both and stages every K / V tile once for the two heads. */
#define TTV_ATTN_GATE 1
#define TTV_ATTN_PAIRED 2
#define TTV_ATTN_QSCALED 4   /* bf16: the q columns already carry the factor head_dim^-0.5 * log2(e) (see ttv_layer_weights.qkv_q_prescaled) */
#define TTV_ATTN_ALLFULL 8   /* the table holds full items only (mode 0 everywhere, padding entries allowed) */
#define TTV_ATTN_PIPE 16     /* bf16, with TTV_ATTN_QSCALED | TTV_ATTN_ALLFULL and no tape: run the software-pipelined kernel (opt-in: measured
                                slower than the plain loop, see ttv_attn.hip; the towers set it under the environment switch TTV_ATTN_PIPE=1) */
#define TTV_ATTN_SPLIT3 32   /* fp32: the split-bf16 ("three-pass") kernel - operands hi + lo in bf16, three bf16 MFMA passes per product, fp32
                                softmax and accumulation (~2^-17 relative per product); the towers set it with ttv_tower_weights.f32_split3 */
#define TTV_ATTN_SPLIT_OUT 64 /* with TTV_ATTN_SPLIT3: the output is written as the split image of the following linear's operand (ttv_split3_pack's
                                format, same bytes as the fp32 output) */
#define TTV_ATTN_SPLIT_IN 128 /* with TTV_ATTN_SPLIT3: q, k and v arrive as planar split images (per 8 features hi0..7 | lo0..7, what the to_qkv linear
                                 of a split tower writes; the gate columns fp32): staged by LDS-DMA */
int ttv_attention(const void* qkvg, int ld, void* out, int ldo, const int32_t* cu_seqlens, const int32_t* qblocks,
                  int n_qblocks, int q_heads, int kv_heads, int head_dim, int flags, int dtype, void* stream);
/* The same operator (transformer.py:100,103) on the 64-query-rows-per-wave kernel: bf16, head_dim 64, q pre-scaled (flags must carry
 * TTV_ATTN_QSCALED; TTV_ATTN_GATE as above).  One workgroup = 4 waves (two workgroups per CU), each wave with 64 query rows (two 32-row tiles that
 * share every K / V fragment read) of any q-head of ONE (sequence, kv-head).  items: device int32 [n_items, 8] =
 * (sequence id, kv-head, wave 0..3: q-head | (first query row / 64) << 8, or -1 for an idle wave, cu_seqlens[sequence], sequence length);
 * sequence id -1 = padding.
 * The host orders the table so that entries i, i+8, ... (one XCD under round-robin dispatch) share a (sequence, kv-head). */
int ttv_attention64(const void* qkvg, int ld, void* out, int ldo, const int32_t* cu_seqlens, const int32_t* items, int n_items,
                    int q_heads, int kv_heads, int head_dim, int flags, int dtype, void* stream);

/* patch_rearrange (model/base/utils.py:26-34) for up to TTV_MAX_CLIPS_PER_LAUNCH clips per call.
 * clips: HOST array of device pointers [n_clips] to [C,T,H,W] tensors; clip_desc: DEVICE int32 [n_clips,8] =
 * (T,H,W, gt,gh,gw, first patch row, C).  Output rows hold the patch vector in (c,pt,ph,pw) order - the
 * reference's (pt,ph,pw,c) order is folded into the packed proj_in / proj_out weight (see weights.py). */
int ttv_patch_gather(const void* const* clips, const int32_t* clip_desc, int clip0, int n_clips, int patch_t, int patch_h,
                     int patch_w, int channels, void* patches, int ld, int dtype, int max_patches_per_clip, void* stream);
/* unpatch_rearrange (model/base/utils.py:37-51); same descriptors, clips are written. */
int ttv_patch_scatter(const void* patches, int ld, const int32_t* clip_desc, int clip0, int n_clips, int patch_t, int patch_h,
                      int patch_w, int channels, void* const* clips, int dtype, int max_patches_per_clip, void* stream);

/* ---- towers (model/base/blocks.py TiTokEncoder 31-104 / TiTokDecoder 108-177) ------------------------- */
typedef struct ttv_tower_dims {
  int32_t kind;          /* TTV_ENCODER / TTV_DECODER */
  int32_t dtype;         /* compute dtype */
  int32_t width;         /* d                                   utils.py:22 */
  int32_t layers;        /*                                     utils.py:9-14 */
  int32_t q_heads, kv_heads, head_dim;   /*                     utils.py:15-20, transformer.py:73-75 */
  int32_t inner;         /* GEGLU hidden I                      transformer.py:39-40 */
  int32_t patch_t, patch_h, patch_w;
  int32_t pix_channels;  /* 3 */
  int32_t token_size;    /* len(fsq_levels) (encoder out / decoder in); 1 for the discriminator use; up to TTV_MAX_TOKEN for the
                            towers in front of / behind ttv_vq_l2_argmin, inference and training alike (the L2 quantiser trains since
                            round 4: ttv_train.hip's check() accepts token_size <= TTV_MAX_TOKEN) */
  float eps;             /* RMSNorm eps 1e-5 */
  float alpha;           /* KEEL residual scale 2*layers        transformer.py:117 */
} ttv_tower_dims;

typedef struct ttv_layer_weights {
  const float* pre_ln;        /* attn_layer.i.pre_ln.weight [d]            */
  const void* to_qkv;         /* attn_layer.i.to_qkv.weight [2d+2g, d]     */
  const void* out_proj;       /* attn_layer.i.out_proj.weight [d, d]       */
  const float* ffd_norm;      /* ffd_layer.i.norm.weight [d]               */
  const void* w12;            /* ffd_layer.i.w12.weight [2I, d]            */
  const void* w3;             /* ffd_layer.i.w3.weight [d, I]              */
  const float* attn_post_ln;  /* attn_post_ln.(i-1).weight, NULL for i==0  */
  const float* ffd_post_ln;   /* ffd_post_ln.(i-1).weight, NULL for i==0   */
  /* optional (bf16, width 256): to_qkv / w12 with the preceding RMSNorm gain folded into the columns
   * (w * gain[None,:]); when non-NULL the pre-norm runs inside the GEMM (rstd from the register-resident row) */
  const void* to_qkv_pn;
  const void* w12_pn;
  /* optional (bf16, width 256): ttv_mlp_pack(w12_pn, w3, out_proj, NEXT layer's to_qkv_pn, rows) - panel images of the fused
   * layer-tail kernel; mlp_pack_qkv_rows = rows of the next layer's to_qkv packed into it (0 = none, e.g. last layer) */
  const void* mlp_pack;
  int32_t mlp_pack_qkv_rows;
  /* 1: the q rows of to_qkv_pn (and of the next-layer image inside the previous layer's mlp_pack) are multiplied by
   * head_dim^-0.5 * log2(e): the projection then emits the softmax exponent directly and the attention kernel runs with
   * TTV_ATTN_QSCALED (one multiply-add less per score).  Inference towers only; `to_qkv` itself is never scaled. */
  int32_t qkv_q_prescaled;
  /* optional (bf16, any width; used where to_qkv_pn is not): an inference copy of to_qkv whose q rows carry the same factor;
   * NULL = use to_qkv and the plain attention kernel */
  const void* to_qkv_qs;
  /* optional mixed bf16 / fp8 linears (BASELINE config #5; bf16 towers, width % 128 == 0, used where the folded width-256 kernels are
   * not): to_qkv (its q rows pre-scaled like to_qkv_qs when that is given) and w12 in OCP e4m3 with one fp32 scale per weight row
   * (ttv_quant_rows_fp8).  When non-NULL the pre-norm output is quantised per token and the two projections run on the fp8 MFMA. */
  const void* to_qkv_f8; const float* to_qkv_f8_scale;
  const void* w12_f8; const float* w12_f8_scale;
  /* optional block-scaled (MX) fp8 linears - ALL FOUR linears of the layer on the fp8 MFMA (round 4): when every pointer below and the
   * four e4m3 images are non-NULL (bf16 towers, width != 256, width % 128 == 0, inner % 128 == 0) the layer quantises each linear's
   * input with ttv_quant_mx_fp8 (one E8M0 scale per 32 consecutive elements) and runs ttv_linear_fp8_mx.  The images then hold
   * ttv_quant_mx_fp8(W', row factors) with W' = to_qkv * pre_ln gain (q rows pre-scaled) / w12 * ffd_norm gain / out_proj / w3: the
   * pre-norm is folded (its rstd is the activation's per-row factor in the epilogue), `*_f8_scale` are the weight rows' fp32 factors,
   * `*_mx` the block scales in ttv_quant_mx_fp8's layout. */
  const void* to_qkv_mx; const void* w12_mx;
  const void* out_proj_f8; const float* out_proj_f8_scale; const void* out_proj_mx;
  const void* w3_f8; const float* w3_f8_scale; const void* w3_mx;
} ttv_layer_weights;

typedef struct ttv_tower_weights {
  const void* proj_in_w;      /* enc: [d, C*pt*ph*pw] columns in (c,pt,ph,pw) order; dec: [d, token_size] */
  const void* proj_in_b;      /* [d] */
  const float* mask_token;    /* [1] */
  const float* ln_pre_t;      /* [d] */
  const float* ln_pre_p;      /* [d] */
  const float* ln_post;       /* [d] */
  const void* proj_out_w;     /* enc: [token_size, d]; dec: [C*pt*ph*pw, d] rows in (c,pt,ph,pw) order */
  const void* proj_out_b;     /* enc: [token_size]; dec: [C*pt*ph*pw] same order */
  const ttv_layer_weights* layers;   /* HOST array [layers] */
  /* optional (decoder, bf16, width 256): proj_out_w * ln_post gain[None,:]; when non-NULL the ln_post RMSNorm runs inside the
   * proj_out GEMM (rows gathered through patch_rows, rstd from the register-resident row) */
  const void* proj_out_pn;
  /* fp32 towers, inference: 1 = "split-bf16" arithmetic (round 4).  proj_in_w and the layers' to_qkv / out_proj / w12 / w3 then point at
   * SPLIT IMAGES made by ttv_split3_pack (same bytes and leading dimension as the fp32 matrix), every linear runs as three bf16 MFMA
   * passes on hi + lo operands (fp32 accumulation) and the attention kernel with TTV_ATTN_SPLIT3; norms, rotary, softmax, GELU, the
   * residual stream, the encoder tail and FSQ stay exact fp32.  Measured: every token index of the reference's fp32 run is kept on the
   * benchmark fixture (max |pre-rounding FSQ value error| ~6e-4) at about a third of the exact-fp32 MFMA kernels' time. */
  int32_t f32_split3;
} ttv_tower_weights;

/* Per-batch metadata, built on the host from Python ints (replaces the device-side bookkeeping and its
 * host syncs at blocks.py:80-88 / 154-162 and rope.py:57-71).  All pointers are DEVICE pointers. */
typedef struct ttv_batch {
  int32_t n_clips;
  int32_t total_rows;          /* L = sum(K_b + P_b) */
  int32_t sum_tokens;          /* sum K_b */
  int32_t sum_patches;         /* sum P_b */
  int32_t max_patches_per_clip;
  int32_t n_qblocks;
  const int32_t* cu_seqlens;   /* [n_clips+1] */
  const int32_t* latent_rows;  /* [sum_tokens]  packed row of every latent token, clip-major */
  const int32_t* patch_rows;   /* [sum_patches] packed row of every patch token, clip-major */
  const int32_t* clip_desc;    /* [n_clips,8] see ttv_patch_gather */
  const int32_t* qblocks;      /* [n_qblocks,4] see ttv_attention (built for this tower's head counts) */
  const float* rope_cs;        /* [L,64] cos|sin, fp64-evaluated on the host as rope.py:48-54 */
  /* training only (may be NULL for inference): */
  const int32_t* blocks64;     /* [n_blocks64,2] (sequence, first row) of every 64-row block (attention backward) */
  const int32_t* row_seq;      /* [L] sequence id of every packed row */
  int32_t n_blocks64;
  int32_t qblocks_paired;      /* 1: entries 2j, 2j+1 of every XCD list of `qblocks` are the same query rows of two q-heads sharing
                                  a kv-head (ttv_attention flag TTV_ATTN_PAIRED); 0: no such guarantee */
  int32_t qblocks_all_full;    /* 1: `qblocks` holds full items only (ttv_attention flag TTV_ATTN_ALLFULL) */
  /* optional: the work table of ttv_attention64 for this tower's head counts (NULL / 0: the towers use `qblocks` only).  When given,
   * inference forwards of bf16 towers whose q columns are pre-scaled run ttv_attention64 instead of ttv_attention. */
  const int32_t* items64;
  int32_t n_items64;
  /* optional: the rotary factors as indices instead of a [L,64] fp32 table (NULL: every kernel reads `rope_cs`).  rope_ids [L,2] int32 =
   * four uint16 per packed row: the position id of axes t, h, w (rope.py:59-67: latent i -> (i,i,i), patch (t,h,w) -> (t,h,w) + K) and
   * the index of the identity row; rope_base fp32 [n_ids + 1, 10, 2] = (cos, sin) of inv_freq[f] * id for every id < n_ids (the values
   * ttv_rope_table_build gathers from) with row n_ids = (1, 0).  8 bytes per row instead of 256: the width-256 to_qkv kernel gathers
   * its factors from the L2-resident base table (bit-identical values). */
  const int32_t* rope_ids;
  const float* rope_base;
  /* optional (NULL / 0: off): an attention work table (format of `qblocks`, full items only) whose entries cover just the query rows of
   * the LATENT tokens of every sequence (rows [0, K_b): the latent tokens come first, blocks.py:85-86).  The encoder's output is read
   * from its latent rows alone (blocks.py:101-103), so with this table ttv_encoder_forward runs the LAST layer's attention for those
   * query rows only (keys / values: every row, as before) and everything behind it - out_proj, KEEL norms, the feed-forward - on the
   * sum K_b latent rows instead of all L: the same values for the rows that are used, the patch rows of the last layer's output are
   * never computed.  Ignored by the decoder and by the training entry points. */
  const int32_t* qblocks_latent;
  int32_t n_qblocks_latent;
  /* optional (NULL / 0: off), the decoder-side twin: an attention work table (format of `qblocks`, full items only) without the query
   * blocks that hold latent rows ONLY (block q of a sequence with (q + 1) * 128 <= K_b).  The decoder's output is read from its patch
   * rows alone (blocks.py:171), so with this table ttv_decoder_forward runs the LAST layer's attention without those blocks: the rows
   * it skips keep the previous layer's attention output, everything behind the attention is row-wise, and no patch row changes a bit.
   * Ignored by the encoder and by the training entry points. */
  const int32_t* qblocks_patch;
  int32_t n_qblocks_patch;
} ttv_batch;

/* Fill ttv_batch.rope_cs [L,64] on the device: rows are gathered from base_cos/base_sin fp32 [n_ids, n_freqs] =
 * cos/sin(inv_freq[f] * n), evaluated once on the host in fp64 exactly as rope.py:40-54 (position ids are small integers,
 * rope.py:59-67: latent i -> (i,i,i); patch (t,h,w) -> (t,h,w) + K).  Replaces RoPE.forward's per-sample loop (rope.py:57-71). */
int ttv_rope_table_build(const float* base_cos, const float* base_sin, int n_ids, int n_freqs, const int32_t* clip_desc,
                         const int32_t* cu_seqlens, const int32_t* row_seq, float* rope_cs, int total_rows, void* stream);

/* bytes of scratch a tower forward needs for this batch */
int64_t ttv_tower_workspace_bytes(const ttv_tower_dims* dims, const ttv_batch* batch);

/* TiTokEncoder.forward (blocks.py:71-104) + FSQ.forward (fsq.py:123-135) fused at the tail:
 * clips (HOST array of device ptrs) -> z [sum_tokens, token_size] fp32 (pre-quantisation, may be NULL),
 * codes [sum_tokens, token_size] (dtype), indices int32 [sum_tokens], bounded fp32 (may be NULL).
 * If fsq == NULL only z is produced (used for the discriminator's encoder, loss_module.py:96-101). */
int ttv_encoder_forward(const ttv_tower_dims* dims, const ttv_tower_weights* w, const ttv_batch* batch,
                        const void* const* clips, const ttv_fsq_params* fsq, float* z, void* codes, int32_t* indices,
                        float* bounded, void* workspace, int64_t workspace_bytes, void* stream);

/* TiTokDecoder.forward (blocks.py:148-177): codes [sum_tokens, token_size] (dtype) -> clips_out
 * (HOST array of device ptrs to [C,T,H,W] buffers, dtype). */
int ttv_decoder_forward(const ttv_tower_dims* dims, const ttv_tower_weights* w, const ttv_batch* batch, const void* codes,
                        void* const* clips_out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- training step: tape-recording forward + backward (reference train.py:65-83 = autograd through the towers) ------- */
/* Transposed linear weights for the data-gradient GEMMs (dX = dY W is run as dY (W^T)^T), compute dtype. */
typedef struct ttv_layer_weights_t {
  const void* to_qkv_t;    /* [d, 2d+2g]  */
  const void* out_proj_t;  /* [d, d]      */
  const void* w12_t;       /* [d, 2I]     */
  const void* w3_t;        /* [I, d]      */
} ttv_layer_weights_t;
typedef struct ttv_tower_weights_t {
  const void* proj_in_t;   /* encoder: [C*pt*ph*pw, d] (packed order); decoder: unused (NULL) */
  const void* proj_out_t;  /* decoder: [d, C*pt*ph*pw] (packed order); encoder: unused (NULL) */
  const ttv_layer_weights_t* layers;   /* HOST array [layers] */
} ttv_tower_weights_t;
/* fp32 gradient buffers, same shapes/layout as the PACKED weights of ttv_tower_weights; zeroed by the caller, accumulated into. */
typedef struct ttv_layer_grads {
  float* pre_ln; float* to_qkv; float* out_proj; float* ffd_norm; float* w12; float* w3; float* attn_post_ln; float* ffd_post_ln;
} ttv_layer_grads;
typedef struct ttv_tower_grads {
  float* proj_in_w; float* proj_in_b; float* mask_token; float* ln_pre_t; float* ln_pre_p; float* ln_post; float* proj_out_w;
  float* proj_out_b;
  const ttv_layer_grads* layers;       /* HOST array [layers] */
  /* optional HOST array [layers] of hipEvent_t (created by the caller): event i is recorded on `stream` right after the last kernel
   * that writes layer i's gradients (the backward visits the layers top down, so these complete in the order layers-1 .. 0).  A
   * data-parallel caller makes its communication stream wait for event i and all-reduces layer i's gradient slice while the
   * backward of the layers below is still running (reference step: train.py:75-83).  NULL = no events. */
  void** layer_done_events;
} ttv_tower_grads;

int64_t ttv_tower_tape_bytes(const ttv_tower_dims* dims, const ttv_batch* batch);
int64_t ttv_tower_bwd_workspace_bytes(const ttv_tower_dims* dims, const ttv_batch* batch);
/* TiTokEncoder.forward recording a tape; z fp32 [sum_tokens, token_size] (FSQ is a separate differentiable op). */
int ttv_encoder_forward_train(const ttv_tower_dims* dims, const ttv_tower_weights* w, const ttv_batch* batch, const void* const* clips,
                              float* z, void* tape, int64_t tape_bytes, void* stream);
/* Backward of the above: dz fp32 -> parameter gradients (accumulated) and, if dclips != NULL, gradients w.r.t. the input
 * clips (HOST array of device ptrs, compute dtype; needed by the discriminator path, loss_module.py:149-152).
 * grads == NULL (then dclips must be given): all parameters frozen - the generator step through the discriminator,
 * loss_module.py:144-151 - only the input gradient is computed, the weight-gradient GEMMs and gain / bias reductions are
 * skipped. */
int ttv_encoder_backward(const ttv_tower_dims* dims, const ttv_tower_weights* w, const ttv_tower_weights_t* wt, const ttv_batch* batch,
                         const float* dz, void* tape, const ttv_tower_grads* grads, void* const* dclips, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* TiTokDecoder.forward recording a tape (workspace >= sum_patches * C*pt*ph*pw elements). */
int ttv_decoder_forward_train(const ttv_tower_dims* dims, const ttv_tower_weights* w, const ttv_batch* batch, const void* codes,
                              void* const* clips_out, void* tape, int64_t tape_bytes, void* workspace, int64_t workspace_bytes,
                              void* stream);
/* Backward: dclips_out (HOST array of device ptrs to d loss / d reconstruction, compute dtype) -> parameter gradients and
 * dcodes fp32 [sum_tokens, token_size] (may be NULL). */
int ttv_decoder_backward(const ttv_tower_dims* dims, const ttv_tower_weights* w, const ttv_tower_weights_t* wt, const ttv_batch* batch,
                         const void* codes, const void* const* dclips_out, void* tape, const ttv_tower_grads* grads, float* dcodes,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* Straight-through FSQ backward (fsq.py:48-51,78-90): dz = dcodes * half_l/half_width * (1 - tanh^2(z + shift)). */
int ttv_fsq_backward(const ttv_fsq_params* p, const float* z, const void* dcodes, int dcodes_dtype, float* dz, int rows, void* stream);

/* Optimizer step of the training loop (reference train.py:76-77 clip_gradients + :183-190 optim.AdamW; the arithmetic of
 * torch.nn.utils.clip_grad_norm_ followed by torch.optim.AdamW, fp32 whatever the tensors' dtype) over a list of tensors, in two launches.
 * table  : device array of n entries {void* param; const void* grad; void* exp_avg; void* exp_avg_sq; int64 numel} (40 bytes each; param,
 *          exp_avg and exp_avg_sq of one dtype - TTV_F32 or TTV_BF16 - which is also the gradient's);
 * chunks : device int32 [n_chunks][2] = (entry index, first element): one block per chunk of up to 8192 elements.
 * ttv_opt_grad_sumsq writes partials[c] = sum of grad^2 over chunk c.  ttv_opt_adamw_step sums partials[0 .. n_partials) in a fixed order
 * (all chunks of ALL tensor lists of the step: the global gradient norm, written to out_norm when not NULL), scales the gradients by
 * min(1, max_norm / (norm + 1e-6)) in registers (max_norm <= 0 or n_partials == 0: no clipping; p.grad is NOT rewritten) and applies
 * AdamW with the given bias corrections 1 - beta1^t and sqrt(1 - beta2^t). */
int ttv_opt_grad_sumsq(const void* table, const int32_t* chunks, int n_chunks, int dtype, float* partials, void* stream);
int ttv_opt_adamw_step(const void* table, const int32_t* chunks, int n_chunks, int dtype, const float* partials, int n_partials, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2_sqrt,
                       float max_norm, float* out_norm, void* stream);

/* Single backward ops, exported for parity tests. */
/* dW[N,K] (fp32, accumulated) += dY[L,N]^T X[L,K]  (weight gradient of y = x w^T; what autograd computes for the
 * nn.Linear weights of base/blocks.py:70-84,147-148).  The token range is split over blocks; with a workspace of
 * ttv_linear_wgrad_workspace_bytes(L, N, K) bytes the split partial tiles are summed in a fixed order (bit-reproducible),
 * with workspace == NULL (workspace_bytes == 0) they are accumulated with fp32 atomics. */
int64_t ttv_linear_wgrad_workspace_bytes(int L, int N, int K);
int ttv_linear_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int L, int N, int K, int dtype,
                     void* workspace, int64_t workspace_bytes, void* stream);
/* Layer-boundary backward in one row-local pass (base/blocks.py:139-168: x1 = attn_post_ln(alpha*x + attn(pre_ln(x))), the
 * same shape around the feed-forward): A = dx + rmsnorm_bwd(x, gain1, dy) joins a sub-layer's pre-norm gradient with the
 * residual gradient; B = rmsnorm_bwd(y, gain2, A) takes it through the KEEL post-norm of the sub-layer below (y = that norm's
 * fp32 input; y == NULL: B = A).  dx (fp32, in/out) = out_scale * B; cast_out (dtype, may be NULL) = B.  dgain1 / dgain2 fp32
 * [width], accumulated, may be NULL. */
int ttv_rmsnorm_backward_chain(const void* x, int ldx, const void* dy, int lddy, const float* gain1, float* dgain1, float* dx, int lddx,
                               const float* y, int ldy, const float* gain2, float* dgain2, float out_scale, void* cast_out, int ldc, int rows,
                               int width, float eps, int dtype, void* stream);
/* RMSNorm backward: dx (dtype), dgain fp32 [width] (accumulated; may be NULL). */
int ttv_rmsnorm_backward(const void* x, int ldx, const void* dy, int lddy, const float* gain, void* dx, int lddx, float* dgain, int rows,
                         int width, float eps, int dtype, void* stream);
/* Attention backward (flash-style recompute from the forward's LSE): qkvg as in ttv_attention, o / dout [L,d], lse fp32
 * [L,q_heads] -> dqkvg [L,2d+2g] (q, k, v column ranges written; gate range untouched).  delta: fp32 [L,q_heads] scratch;
 * dkv_scratch: fp32 [L,2g] scratch (fp32 dtype only). blocks64 / row_seq as in ttv_batch.  rope_cs (fp32 [L,64] cos|sin as in
 * ttv_batch, may be NULL): when given, dq and dk are returned as gradients w.r.t. the q / k BEFORE the rotary embedding
 * (rope.py:19-27), i.e. the transposed rotation is applied in fp32 before the store. */
int ttv_attention_backward(const void* qkvg, int ld, const void* o, int ldo, const void* dout, int ldd, const float* lse, float* delta,
                           const int32_t* cu_seqlens, const int32_t* blocks64, int n_blocks64, const int32_t* row_seq, void* dqkvg,
                           int ldg, float* dkv_scratch, int total_rows, int q_heads, int kv_heads, int dtype, const float* rope_cs,
                           void* stream);
/* ttv_attention with an extra fp32 [L,q_heads] log-sum-exp output (training forward). */
int ttv_attention_lse(const void* qkvg, int ld, void* out, int ldo, const int32_t* cu_seqlens, const int32_t* qblocks, int n_qblocks,
                      int q_heads, int kv_heads, int head_dim, int flags, int dtype, float* lse, void* stream);

/* ---- codebook statistics (train_utils/codebook_logging.py:19-32) -------------------------------------- */
/* counts[idx] += 1 for every index (int64 device histogram, atomics); usage/entropy are finished on the host. */
int ttv_codebook_histogram(const int32_t* indices, int n, int64_t* counts, int codebook_size, void* stream);

/* L1 reconstruction term of the generator loss (loss_module.py:118 per clip, mean over clips - train.py:70), value and
 * gradient in one launch: *loss += mean_c mean_i |recon_c[i] - target_c[i]| (caller zeroes it);
 * grad_c[i] = sign(recon - target) / (sizes[c] * n_clips) in `dtype` (grad NULL = value only).  Host arrays of device pointers. */
int ttv_l1_loss(void* const* recon, void* const* target, void* const* grad, const int32_t* sizes, int n_clips, int dtype, float* loss,
                void* stream);

/* The loader's tail on the device (dataset/video_dataset.py:116-119: ToDtype(scale=True) + Normalize(0.5, 0.5) on channel-first
 * frames): decoded frames uint8 [T,H,W,3] (device memory, what the decoder / a shard hands over) -> clip [3,T,H,W] in `dtype`,
 * value u8 / 127.5 - 1 evaluated in fp32 and rounded once.  T*H*W % 4 == 0 (patch-aligned clips always are). */
int ttv_clip_from_u8(const void* frames_thwc, int T, int H, int W, void* clip_cthw, int dtype, void* stream);

/* PSNR statistic of the evaluation loop (model/metrics/eval_metrics.py:19,32-36: x.clamp(-1, 1), torchmetrics
 * PeakSignalNoiseRatio(data_range=2) = running sum of squared errors + element count): acc[0] += sum (clamp(recon) - target)^2,
 * acc[1] += number of elements, both double, device memory, over the clips of the call (host arrays of device pointers, `dtype`).
 * PSNR = 10 log10(4 * acc[1] / acc[0]) is finished on the host (one read when the score is wanted, none per step). */
int ttv_sq_err_accumulate(void* const* recon, void* const* target, const int32_t* sizes, int n_clips, int dtype, int clamp, double* acc,
                          void* stream);

/* ---- measurement hook (bench.py roofline leg) ---------------------------------------------------------- */
/* Kernel classes whose launches can be bracketed by HIP events on the stream they are launched on. */
#define TTV_KC_ATTENTION 1
#define TTV_KC_GEMM_QKV 2
#define TTV_KC_GEMM_GEGLU 3
#define TTV_KC_GEMM_RESID 4
#define TTV_KC_GEMM_STORE 5
#define TTV_KC_RMSNORM 6
#define TTV_KC_PATCH 7
#define TTV_KC_ROWS 8
/* Start recording launches of `kernel_class` (up to max_records; events are created here, outside any launch path).
 * This is the only process-global state in the library and it is off by default. */
int ttv_prof_begin(int kernel_class, int max_records);
/* Diagnostics for kernel ablation timing / tests (never set in product use): bit0 = GEMM epilogues skip their stores;
 * bit7 (128) / bit8 (256) = force the 160- / 128-token tile of the general-K GEMM instead of the grid-balance choice.
 * The flags belong to the CALLING HOST THREAD (thread-local): launches made by other threads are unaffected. */
int ttv_debug_set(int flags);
/* Diagnostics: device buffer (>= 256 int64) that instrumented kernels fill with s_memtime stamps of block 0; NULL = off. */
int ttv_debug_stamps(void* device_buffer);
/* Synchronise the recorded events, return their summed duration (ms) and count, and release them. */
int ttv_prof_end(double* total_ms, int* count);

#ifdef __cplusplus
}
#endif
#endif /* TITOK_HIP_H */

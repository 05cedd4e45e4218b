// Internal launch functions (host side).  The C-ABI in include/titok_hip.h is layered on these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/titok_hip.h"

// ---- ttv_elem.hip ----
int ttvk_rmsnorm(const void* in, int in_dtype, int ld_in, const int* src_rows, void* out, int out_dtype, int ld_out,
                 const int* dst_rows, const float* gain, int rows, int d, float eps, hipStream_t s, float* next_rstd = nullptr,
                 void* mx_q = nullptr, void* mx_s = nullptr,    // mx_q / mx_s: the stored row also as block-scaled e4m3 [rows, d] + its E8M0 scales
                 int split_image = 0);                          // fp32 output written as the split-bf16 image (hi0..3 | lo0..3 per four elements)
int ttvk_row_rstd(const void* in, int dtype, int ld_in, float* rstd, int rows, int d, float eps, hipStream_t s);
// dst[dst_rows ? dst_rows[r] : r] = src[src_rows ? src_rows[r] : r] for r < rows; row_bytes a multiple of 16, rows 16-byte aligned
int ttvk_copy_rows(const void* src, int64_t ld_src_bytes, const int* src_rows, void* dst, int64_t ld_dst_bytes, const int* dst_rows, int rows,
                   int row_bytes, hipStream_t s);
int ttvk_fill_const_rows(void* x, int dtype, int ld, const int* rows_map, int rows, int d, const float* mask_token,
                         const float* gain, float eps, hipStream_t s);
int ttvk_dec_embed(const void* codes, int C, const void* w, const void* bias, const float* mask_token, const float* gain,
                   void* x, int dtype, int ld, const int* rows_map, int rows, int d, float eps, hipStream_t s);
int ttvk_fsq_forward(const ttv_fsq_params* p, const void* z, int z_dtype, int rows, void* codes, int codes_dtype, int* indices,
                     float* bounded, hipStream_t s);
int ttvk_fsq_indices_to_codes(const ttv_fsq_params* p, const int* indices, int rows, void* codes, int codes_dtype, hipStream_t s);
int ttvk_enc_tail(const void* x, int dtype, int ld, const int* rows_map, int rows, int d, const float* gain, float eps,
                  const void* w, const void* bias, int C, const ttv_fsq_params* fsq, float* z_out, void* codes, int* indices,
                  float* bounded, hipStream_t s);
int ttvk_patch_copy(bool scatter, void* const* clips, const int* clip_desc, int clip0, int n_clips, int pt, int ph, int pw,
                    int C, void* patches, int ld, int dtype, int max_patches, hipStream_t s);
int ttvk_rope_apply(void* x, int dtype, int ld, int rows, int heads, const float* cs, hipStream_t s);
int ttvk_histogram(const int* idx, int n, int64_t* counts, int size, hipStream_t s);
int ttvk_quant_rows_fp8(const void* in, int in_dtype, int ld_in, const float* gain, float eps, void* out, int ld_out, float* scales, int rows,
                        int d, hipStream_t s);

int ttvk_split3_pack(const float* w, int ldw, void* out, int ldo, int rows, int K, hipStream_t s);
int64_t ttvk_mx_scale_ld(int d);     // bytes of E8M0 scales per row of width d
int ttvk_quant_mx_fp8(const void* in, int in_dtype, int ld_in, void* out, int ld_out, void* mx, float* row_scales, int rows, int d, hipStream_t s);

int ttvk_clip_from_u8(const void* frames, long long n_pix, void* clip, int dtype, hipStream_t s);

// ---- ttv_gemm.hip ----
// out^T-oriented GEMM: the MFMA "row" side is the output feature (W rows), the "column" side the token (X rows),
// so every lane owns 4 consecutive output features of one token and the epilogues below are lane-local.
enum GemmEpilogue {
  EPI_STORE = 0,   // y = acc (+bias) (+*add_scalar), stored in dtype
  EPI_QKV_ROPE,    // y = acc with rotary applied to the q and k column ranges (transformer.py:87,97-98)
  EPI_GEGLU,       // y[:, f] = gelu(acc[:, I+f]) * acc[:, f]           (transformer.py:51-52)
  EPI_RESID_T,     // y = alpha*resid + acc, stored in dtype            (layer 0 residual, transformer.py:129-130)
  EPI_RESID_F32,   // y = alpha*resid + acc, stored fp32                (KEEL pre-post-norm sum, transformer.py:141,144)
  EPI_RESID_NORM,  // y = RMSNorm(alpha*resid + acc) * norm_gain, stored in dtype (whole KEEL step, transformer.py:141-145);
                   // needs a kernel whose waves own full rows: bf16, K == 256, N == 256 (ttvk_gemm_supports_resid_norm)
  EPI_STORE_PATCH, // y = acc + bias written straight into the clips as patches (blocks.py:173-176 + utils.py:37-51; bf16, K = 256)
};

struct GemmArgs {
  const void* x; int ldx;          // tokens  [M,K]
  const void* w; int ldw;          // weights [N,K]  (EPI_GEGLU: [2I,K], N = I)
  int M, N, K;
  void* y; int ldy;
  const void* bias;                // [N] dtype or null
  const float* add_scalar;         // device scalar or null
  const void* resid; int ldr;      // [M,N] dtype
  float alpha;
  const float* rope_cs;            // [M,64]
  const int* rope_ids; const float* rope_base;   // optional (ttv_batch.rope_ids / rope_base): the K == 256 bf16 kernel gathers its factors through them
  int rope_q_end, rope_k_begin, rope_k_end;  // column ranges [0,q_end) and [k_begin,k_end) get rotary
  int dtype;
  const float* norm_gain;          // EPI_RESID_NORM: post-norm gain [N]
  // EPI_RESID_NORM, optional (training tape): the pre-norm sum alpha * resid + acc in fp32, and RMSNorm(y) * norm_gain2 of the stored
  // (rounded) row = the next pre-norm's output, so that neither needs a launch of its own
  float* sum_f32; int ld_sum;
  void* y2; int ldy2; const float* norm_gain2;
  void* yq; void* yq_mx;           // ttvk_gemm_fp8 with block scales, EPI_GEGLU: the output leaves as block-scaled e4m3 [M, N] (ld N) + E8M0 scales
                                   // (k_quant_mx_fp8's layout for width N) instead of bf16 y - the next linear's operand, no pass of its own
  int split3;                      // fp32 only: w is the split-bf16 image of the weight (ttv_split3_pack) and the products run as three bf16
                                   // MFMA passes (k_gemm_f32<.., SPLIT>)
  int x_image, y_image;            // split3: x is already a split image (written by its producer: no split in the staging) / y is written as one
                                   // (1: per four features hi0..3 | lo0..3, STORE / GEGLU; 2: to_qkv - q, k, v per eight features hi0..7 | lo0..7,
                                   // the attention kernel's operand format, gate columns fp32)
  int prenorm;                     // 1: w has the RMSNorm gain folded in, x is the un-normalised row (bf16, K == 256 only)
  const float* row_scale;          // optional [M]: output row t is multiplied by row_scale[t] before the epilogue (the rstd of a pre-norm
                                   // whose gain is folded into w: any K; bf16 kernels)
  const int* x_rows;               // optional (bf16, K == 256): GEMM row t reads x row x_rows[t]
  float eps;
  // EPI_STORE_PATCH (output side) / gather (EPI_STORE, input side): row t of the GEMM is patch t (clip-major); its clip is
  // row_seq[patch_rows[t]]; x (gather) resp. y (scatter) is then unused
  int gather;
  void* const* clips;              // HOST array of device pointers to the [C,T,H,W] outputs of clips clip0 .. clip0+n_clips-1
  int n_clips;
  const int* clip_desc;            // device [*,8], see ttv_patch_gather
  const int* patch_rows;           // device [M]
  const int* row_seq;              // device [L]
  int patch_t, patch_h, patch_w;   // powers of two; patch_w * sizeof(bf16) == 16
};
struct ClipPtrs { void* p[TTV_MAX_CLIPS_PER_LAUNCH]; };
int ttvk_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s);
// x_mx / w_mx: E8M0 block scales (ttvk_quant_mx_fp8's layout) - both or neither; with them the fp32 row factors are optional
int ttvk_gemm_fp8(GemmEpilogue epi, const GemmArgs& a, const float* x_scale, const float* w_scale, hipStream_t s, const void* x_mx = nullptr,
                  const void* w_mx = nullptr);
bool ttvk_gemm_supports_resid_norm(int dtype, int N, int K);

// ---- ttv_attn.hip ----
int ttvk_attention(const void* qkvg, int ld, void* out, int ldo, const int* cu_seqlens, const int* qblocks, int n_qblocks,
                   int q_heads, int kv_heads, int head_dim, int flags, int dtype, hipStream_t s, float* lse_out = nullptr,
                   void* out_raw = nullptr);

// bf16 tables of full items, pre-scaled q, no tape outputs: the software-pipelined kernel (ttv_attn_swp.hip); ttvk_attention dispatches
int ttvk_attention_swp(const void* qkvg, int ld, void* out, int ldo, const int* cu_seqlens, const int* qblocks, int n_qblocks,
                       int q_heads, int kv_heads, int gate_mul, hipStream_t s);
int ttvk_attention_mxout(const void* qkvg, int ld, void* out_q, void* out_mx, int ld_mx, const int* cu_seqlens, const int* qblocks,
                         int n_qblocks, int q_heads, int kv_heads, hipStream_t s);

// ---- ttv_attn64.hip ----
int ttvk_attention64(const void* qkvg, int ld, void* out, int ldo, const int* cu_seqlens, const int* items, int n_items, int q_heads,
                     int kv_heads, int flags, hipStream_t s);

// ---- ttv_vq.hip (nearest-codebook-entry quantiser) ----
int ttvk_vq_norms(const void* cb, int dtype, int ld, int N, int C, float* cnorm, hipStream_t s);
int64_t ttvk_vq_workspace_bytes(int rows);
int ttvk_vq_l2_argmin(const void* z, int dtype, int ldz, const void* cb, int ldc, const float* cnorm, int rows, int N, int C, int* indices,
                      float* best_dist, void* workspace, int64_t workspace_bytes, hipStream_t s);
int ttvk_vq_lookup(const void* cb, int dtype, int ldc, const int* indices, int rows, int C, void* codes, int ldo, hipStream_t s);
int ttvk_vq_lookup_bwd(const void* dcodes, int dtype, int ld, const int* indices, int rows, int C, float* dcb, int ldc, hipStream_t s);

// ---- ttv_mlp.hip ----
bool ttvk_mlp_fused_supported(int dtype, int width, int inner);
struct MlpNextQkv {     // optional fused back of the layer-tail kernel: the next layer's qkv projection + rotary
  void* qkv; int ld;    // [M, ld] output, (q | gate | k | v)
  const float* rope_cs; // [M, 64] (cos | sin)
  int rows;             // rows of the folded to_qkv packed behind the feed-forward images (% 64 == 0)
  int rope_q_end, rope_k_begin, rope_k_end;
};
int64_t ttvk_mlp_pack_bytes(int inner, int next_qkv_rows);
int ttvk_mlp_pack(const void* w12_folded, const void* w3, const void* wo, const void* next_qkv_folded, int next_qkv_rows, int inner,
                  void* packed, hipStream_t s);
int ttvk_mlp_fused(const void* ao, int ldao, const float* front_gain, float front_alpha, const void* x, int ldx, const void* packed,
                   int inner, void* y, int ldy, const float* post_gain, float alpha, float eps, int M, const MlpNextQkv* nq, hipStream_t s);

// ---- ttv_bwd.hip (backward kernels) ----
int ttvk_l1_loss(void* const* recon, void* const* target, void* const* grad, const int* sizes, int n_clips, int total_clips, int dtype,
                 float* loss, hipStream_t s);
int ttvk_sq_err(void* const* recon, void* const* target, const int* sizes, int n_clips, int dtype, int clamp, double* acc2, hipStream_t s);
// dx (fp32, in/out) = out_scale * B(A), cast_out = (T) B(A) with A = dx + rmsnorm_bwd(x, gain1, dy), B = rmsnorm_bwd(y, gain2, .) or
// identity when y == NULL; gain gradients accumulated with atomics (dgain1 / dgain2 may be NULL); cast_out may be NULL
int ttvk_rmsnorm_bwd_chain(const void* x, int ldx, const void* dy, int lddy, const float* gain1, float* dgain1, float* dx, int lddx,
                           const void* y, int y_dt, int ldy, const float* gain2, float* dgain2, float out_scale, void* cast_out, int ldc, int rows,
                           int d, float eps, int dt, hipStream_t s);
int ttvk_rmsnorm_bwd(const void* x, int x_dt, int ldx, const int* xr, const void* dy, int dy_dt, int lddy, const int* dyr,
                     const float* gain, void* dx, int dx_dt, int lddx, const int* dxr, int acc, float* dgain, int rows, int d, float eps,
                     hipStream_t s);
int ttvk_colsum(const void* a, int dt, int lda, const int* rows_map, int rows, int n, float* out, hipStream_t s);
int ttvk_sumall(const void* a, int dt, int lda, const int* rows_map, int rows, int n, const float* colw, float scale, float* out, hipStream_t s);
int ttvk_gate_fwd(const void* a, int lda, const void* gate, int ldg, void* ag, int ldo, int rows, int d, int dt, hipStream_t s);
int ttvk_gate_bwd(const void* dag, int ldd, const void* a, int lda, const void* gate, int ldg, void* da, int ldda, void* dgate, int lddg,
                  int rows, int d, int dt, float* delta, hipStream_t s);
int ttvk_geglu_fwd(const void* u, int ldu, void* h, int ldh, int rows, int I, int dt, hipStream_t s);
int ttvk_geglu_bwd(const void* u, int ldu, const void* dh, int lddh, void* du, int lddu, int rows, int I, int dt, hipStream_t s);
int ttvk_scale_cast(const float* a, float alpha, float* b, void* c, int dt, long n, hipStream_t s);
int ttvk_to_f32(const void* a, int dt, float* b, long n, int accumulate, hipStream_t s);
int ttvk_fsq_bwd(const ttv_fsq_params* fp, const float* z, const void* dcodes, int dt, float* dz, int rows, hipStream_t s);
// part / part_bytes: optional scratch for the split partial tiles (ttvk_wgrad_ws_bytes); without it the bf16 path uses fp32 atomics
// `batch` (optional): the split partial tiles of several weight gradients are summed by ONE launch (ttvk_wgrad_flush) instead of one
// k_wgrad_reduce per weight: the call takes its partial tiles from part + batch->used_bytes, records what to sum and returns; the
// gradients are complete only behind the flush.  A call that does not fit (more than TTV_WGRAD_BATCH entries, scratch exhausted,
// or a shape that takes another path) flushes what is pending and proceeds on its own.
#define TTV_WGRAD_BATCH 6
struct WgradBatch {
  int n = 0;
  int64_t used_bytes = 0;
  struct Entry { const float* part; float* dw; int splits, lddw, N, K, tiles_n, tiles; } e[TTV_WGRAD_BATCH];
};
int ttvk_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int L, int N, int K, int dt, float* part,
               int64_t part_bytes, hipStream_t s, WgradBatch* batch = nullptr);
int ttvk_wgrad_flush(WgradBatch* batch, hipStream_t s);
int64_t ttvk_wgrad_ws_bytes(int L, int N, int K);
int ttvk_outer_small(const void* a, int a_dt, int lda, int C, const void* b, int b_dt, int ldb, const int* b_rows, float* dw, int lddw,
                     int transpose_out, int rows, int d, hipStream_t s);
int ttvk_expand_small(const void* a, int a_dt, int lda, int C, const void* w, int w_dt, int ldw, int w_cf, void* out, int o_dt, int ldo,
                      int rows, int d, hipStream_t s);
int ttvk_reduce_small(const void* a, int a_dt, int lda, const int* a_rows, const void* w, int w_dt, int ldw, int C, float* out, int ldo,
                      int rows, int d, hipStream_t s);
int ttvk_attention_bwd(const void* qkvg, int ld, const void* o, int ldo, const void* dout, int ldd, const float* lse, float* delta,
                       const int* cu, const int* blocks64, int n_blocks64, const int* row_seq, void* dqkvg, int ldg, float* dkv_scratch,
                       int total_rows, int hq, int hkv, int dt, const float* rope_cs, hipStream_t s, int delta_ready = 0, const int* qlim_desc = nullptr);
int ttvk_rope_apply_dir(void* x, int dtype, int ld, int rows, int heads, const float* cs, int conj, hipStream_t s);
int ttvk_dec_embed_ex(const void* codes, int C, const void* w, const void* bias, const float* mask_token, const float* gain, void* x,
                      int dtype, int ld, const int* rows_map, int rows, int d, float eps, void* hpre, hipStream_t s);
int ttvk_const_rows_bwd(const float* colsum, const float* mask_token, const float* gain, int dt, float eps, float* dgain, float* dmask,
                        int d, hipStream_t s);
int ttvk_rope_build(const float* base_cos, const float* base_sin, int n_ids, int F, const int* clip_desc, const int* cu, const int* row_seq,
                    float* out, int total_rows, hipStream_t s);

// Linear layers of the TiTok-Video towers (proj_in/out, to_qkv, out_proj, w12, w3) as y = x @ w^T.
//
// Orientation: the MFMA "row" (A) operand is the WEIGHT tile (output features), the "column" (B) operand the
// TOKEN tile, i.e. the kernels compute y^T = w @ x^T.  With mfma_f32_16x16x32_bf16's C layout
// (col = lane&15, row = 4*(lane>>4)+reg) a lane then owns 4 CONSECUTIVE output features of one token, so
// rotary pairs, GEGLU's (x, gate) pair, bias and the residual are lane-local.  Both operands are K-contiguous in
// memory ([N,K] weights, [M,K] activations), which is what MFMA wants.
//
// Epilogue (epilogue_tile): all side loads of a wave's tile (rotary table / residual / bias) are issued first,
// then the math, then the stores - no per-element branch, so nothing serialises on a memory round trip.  bf16 rows
// are written as 16-byte vectors: lanes kq and kq^1 (lane ^ 16) swap one 4-feature group so that each holds 8
// consecutive features (64 contiguous bytes per token per store instruction).
//
// Kernels
//   k_gemm_k256 : K == 256 (every to_qkv / w12 / out_proj / decoder proj_out of a width-256 tower).  These GEMMs
//                 are HBM-bound (192 FLOP/B at N=768), so the structure minimises traffic and exposed latency:
//                 a wave keeps its 32 tokens x K in registers (loaded once per 128-token tile), 64-row weight
//                 panels stream through double-buffered LDS, blocks are persistent over contiguous
//                 (token tile, panel) items, and the preceding RMSNorm is folded in (PRENORM).
//   k_gemm_bf16 : general K: 128 features x 128 tokens x 64 K tiles, 4 waves (2x2), register-staged double-buffered
//                 LDS, XOR-swizzled 16-byte chunks (conflict-free ds_read_b128), XCD-aware tile order.
//   k_gemm_f32  : fp32 towers (bit-exact token indices): exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), 128x128x32 tiles.
#include <stdlib.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

struct GemmDev {
  const void* x; const void* w; void* y;
  const void* bias; const float* add_scalar; const void* resid; const float* rope_cs;
  int ldx, ldw, ldy, ldr;
  int M, N, K;
  int w_rows;  // rows of w that may be read (N, or 2I for GEGLU)
  float alpha;
  int rope_q_end, rope_k_begin, rope_k_end;
  float eps;  // RMSNorm eps for the folded pre-norm (k256 kernel)
  int debug;  // diagnostics (ttv_debug_set): bit0 = skip epilogue stores
  const float* norm_gain;
  // EPI_STORE_PATCH (decoder tail): GEMM row t is patch t; output goes into the clip tensors as patches
  ClipPtrs clips;
  const int* clip_desc; const int* patch_rows; const int* row_seq;
  const int* x_rows;   // k256: GEMM row t reads x row x_rows[t] (NULL = identity)
  const float* row_scale;   // optional [M]: acc rows are multiplied by it first (folded pre-norm of the generic-K kernels)
  const int* rope_ids; const float* rope_base;   // optional: rotary factors by position id (k256 QKV kernel; see ttv_batch.rope_ids)
  const float* x_scale; const float* w_scale;   // fp8 operands: per-token / per-weight-row dequantisation factors (k_gemm_fp8_dma; either may be NULL with MX)
  // EPI_RESID_NORM extras (training tape, round 4): the pre-norm sum in fp32, and a SECOND RMSNorm of the row just produced
  float* sum_f32; int ld_sum;                   // optional [M, N] fp32: alpha * resid + acc (what the post-norm's backward needs)
  void* y2; int ldy2; const float* norm_gain2;  // optional [M, N] dtype: RMSNorm(y) * norm_gain2 with y as stored (rounded): the next pre-norm's output
  int split3;                                   // fp32 kernel: w is the split-bf16 image (hi | lo per 16-byte chunk), products in three bf16 passes
  int x_image, y_image;                         // split3: x already IS a split image (its producer wrote it) / the fp32 output is written as one
  const uint8_t* x_mx; const uint8_t* w_mx;     // MX block scales (E8M0, k_quant_mx_fp8's layout), ld_mx bytes per row
  int ld_mx;
  uint8_t* yq; uint8_t* yq_mx; int ld_yq_mx, yq_nkp;   // EPI_GEGLU of the MX kernel: h leaves as block-scaled e4m3 [M, N] + scales instead of bf16
  int clip0, pt_shift, ph_shift;   // log2(patch_t), log2(patch_h); patch_w == 8
  long long* stamps;               // diagnostic builds only (-DQKV_STAMPS): g_ttv_stamps
};

// per-token part of a patch destination (EPI_STORE_PATCH), computed once per token tile
struct PatchDst {
  bf16_t* base;        // clip + first pixel of the patch (channel 0, ipt = iph = 0)
  int thw, hw, w;      // channel / frame / image-row strides of that clip, in elements
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// split-bf16 towers: erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7 - two orders below the 2^-17 of the three-pass products;
// the exact-fp32 towers keep libm's erff): one v_rcp, one v_exp and a degree-5 Horner chain instead of ~50 branchy instructions
__device__ __forceinline__ float gelu_erf_as(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(x * x * -1.44269504088896340736f);
  const float erf_abs = fmaf(-poly, e, 1.0f);
  return 0.5f * v * (1.0f + __builtin_copysignf(erf_abs, v));
}

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below bf16 resolution) for the bf16 GEGLU epilogue:
// one v_rcp, one v_exp and a degree-5 Horner chain instead of libm's branchy erff.
__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
  bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  return __builtin_bit_cast(uint2, b);
}

// split image of four fp32 values (one 16-byte chunk): (hi0..3 | lo0..3), hi = bf16(x) (round to nearest even), lo = bf16(x - hi)
__device__ __forceinline__ uint4 split4_bf16(uint4 v) {
  const float x0 = __uint_as_float(v.x), x1 = __uint_as_float(v.y), x2 = __uint_as_float(v.z), x3 = __uint_as_float(v.w);
  const bf16_t h0 = (bf16_t)x0, h1 = (bf16_t)x1, h2 = (bf16_t)x2, h3 = (bf16_t)x3;       // round to nearest even (v_cvt_pk_bf16_f32)
  const bf16_t l0 = (bf16_t)(x0 - (float)h0), l1 = (bf16_t)(x1 - (float)h1), l2 = (bf16_t)(x2 - (float)h2), l3 = (bf16_t)(x3 - (float)h3);
  const bf16x4 hv = {h0, h1, h2, h3}, lv = {l0, l1, l2, l3};
  const uint2 hp = __builtin_bit_cast(uint2, hv), lp = __builtin_bit_cast(uint2, lv);
  return make_uint4(hp.x, hp.y, lp.x, lp.y);
}

// ------------------------------------------------------------------------------------------------
// Wave-tile epilogue.  acc[i][j]: output m-tile i (features feat[i] .. feat[i]+3 for this lane), n-tile j (token tok[j]).
// acc2 = gate half for EPI_GEGLU.  kq = lane >> 4.  T = storage type of x / y / resid.
// ------------------------------------------------------------------------------------------------
template <int EPI, typename T, int NI, int NJ>
__device__ __forceinline__ void epilogue_tile(const GemmDev& p, const int (&tok)[NJ], const int (&feat)[NI],
                                              f32x4 (&acc)[NI][NJ], f32x4 (&acc2)[NI][NJ], int kq, const PatchDst* pd = nullptr) {
  if (p.debug & 1) {  // timing-only path: keep the values alive, store nothing
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][3]), "v"(acc2[i][j][1]));
    return;
  }
  int tc[NJ], fc[NI];
  bool tv[NJ], fv[NI];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { tv[j] = tok[j] < p.M; tc[j] = tv[j] ? tok[j] : p.M - 1; }
#pragma unroll
  for (int i = 0; i < NI; ++i) { fv[i] = feat[i] < p.N; fc[i] = fv[i] ? feat[i] : p.N - 4; }

  if (p.row_scale) {     // folded pre-norm: y = rstd[token] * (x (W o gain)^T); uniform branch
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float rs = p.row_scale[tc[j]];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i][j] *= rs;
        if (EPI == EPI_GEGLU) acc2[i][j] *= rs;
      }
    }
  }

  if (EPI == EPI_STORE || EPI == EPI_STORE_PATCH) {
    if (p.bias) {
      f32x4 b[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) b[i] = Vec4<T>::load((const T*)p.bias + fc[i]);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] += b[i];
    }
    if (p.add_scalar) {
      const float sc = round_to<T>(p.add_scalar[0]);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (p.bias) {  // the Linear output is rounded to dtype before the scalar is added (blocks.py:97)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = round_to<T>(acc[i][j][e]);
          }
          acc[i][j] += sc;
        }
    }
  } else if (EPI == EPI_QKV_ROPE) {
    // the rotary column ranges are multiples of the block's feature tile (checked on the host): uniform per block
    const int f_first = __builtin_amdgcn_readfirstlane(feat[0]);
    if (f_first < p.rope_q_end || (f_first >= p.rope_k_begin && f_first < p.rope_k_end)) {
      float2 c[NI][NJ], s[NI][NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float* cs = p.rope_cs + (size_t)tc[j] * 64 + ((fc[i] & 63) >> 1);
          c[i][j] = *reinterpret_cast<const float2*>(cs);
          s[i][j] = *reinterpret_cast<const float2*>(cs + 32);
        }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4 a = acc[i][j];
          acc[i][j] = (f32x4){a[0] * c[i][j].x - a[1] * s[i][j].x, a[0] * s[i][j].x + a[1] * c[i][j].x,
                              a[2] * c[i][j].y - a[3] * s[i][j].y, a[2] * s[i][j].y + a[3] * c[i][j].y};
        }
    }
  } else if (EPI == EPI_GEGLU) {
    if (p.resid) {
      // training tape: the pre-activations u = [x | gate] (ld = ldr) are kept as well; h is then formed from the STORED (rounded)
      // values, exactly what a separate GEGLU pass over u would see
      T* u = (T*)const_cast<void*>(p.resid);
      if (sizeof(T) == 2 && (NI % 2 == 0) && p.N % 8 == 0 && p.ldr % 8 == 0) {
        // 16-byte stores as at the end of this function: lanes kq / kq ^ 1 exchange one 4-feature group of each half (round 5; 8-byte before)
        const bool odd = kq & 1;
#pragma unroll
        for (int ip = 0; ip < NI / 2; ++ip) {
          const int i0 = 2 * ip, i1 = 2 * ip + 1;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const uint4 ox = xchg16_pair(pack_bf16x4(acc[i0][j]), pack_bf16x4(acc[i1][j]));
            const uint4 og = xchg16_pair(pack_bf16x4(acc2[i0][j]), pack_bf16x4(acc2[i1][j]));
            const int start = odd ? feat[i1] - 4 : feat[i0];
            if (tv[j] && start + 8 <= p.N) {
              *reinterpret_cast<uint4*>(u + (size_t)tok[j] * p.ldr + start) = ox;
              *reinterpret_cast<uint4*>(u + (size_t)tok[j] * p.ldr + p.N + start) = og;
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            if (tv[j] && fv[i]) {
              Vec4<T>::store(u + (size_t)tok[j] * p.ldr + feat[i], acc[i][j]);
              Vec4<T>::store(u + (size_t)tok[j] * p.ldr + p.N + feat[i], acc2[i][j]);
            }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[i][j][e] = round_to<T>(acc[i][j][e]); acc2[i][j][e] = round_to<T>(acc2[i][j][e]); }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = geglu_fast(acc2[i][j][e], acc[i][j][e]);
        } else if (p.split3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = gelu_erf_as(acc2[i][j][e]) * acc[i][j][e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = gelu_erf(acc2[i][j][e]) * acc[i][j][e];
        }
  } else {  // EPI_RESID_T / EPI_RESID_F32
    f32x4 r[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) r[i][j] = Vec4<T>::load((const T*)p.resid + (size_t)tc[j] * p.ldr + fc[i]);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] += p.alpha * r[i][j];
  }

  // ---- stores ----
  if (EPI == EPI_RESID_F32) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (tv[j] && fv[i]) *reinterpret_cast<f32x4*>((float*)p.y + (size_t)tok[j] * p.ldy + feat[i]) = acc[i][j];
  } else if (sizeof(T) == 2 && (NI % 2 == 0)) {
    // lanes kq / kq^1 exchange one 4-feature group: even kq keeps m-tile i0 (8 consecutive features from feat[i0]),
    // odd kq keeps m-tile i1 (8 consecutive features from feat[i1]-4)
    const bool odd = kq & 1;
#pragma unroll
    for (int ip = 0; ip < NI / 2; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const uint2 p0 = pack_bf16x4(acc[i0][j]), p1 = pack_bf16x4(acc[i1][j]);
        const uint4 out = xchg16_pair(p0, p1);     // v_permlane16_swap: even kq (own p0, partner's p0), odd kq (partner's p1, own p1)
        const int start = odd ? feat[i1] - 4 : feat[i0];
        if (EPI == EPI_STORE_PATCH) {
          // 8 consecutive features = the pw = 8 pixels of one (c, ipt, iph) image row of the patch (utils.py:37-51, patch
          // vector order (c, pt, ph, pw)): one 16-byte store into the clip
          const int sg = start >> 3;
          const int c = sg >> (p.pt_shift + p.ph_shift), ipt = (sg >> p.ph_shift) & ((1 << p.pt_shift) - 1), iph = sg & ((1 << p.ph_shift) - 1);
          if (tv[j] && start + 8 <= p.N) *reinterpret_cast<uint4*>(pd[j].base + c * pd[j].thw + ipt * pd[j].hw + iph * pd[j].w) = out;
        } else if (tv[j] && start + 8 <= p.N) {
          *reinterpret_cast<uint4*>((T*)p.y + (size_t)tok[j] * p.ldy + start) = out;
        }
      }
    }
  } else if (sizeof(T) == 4 && p.y_image == 2) {
    // split-bf16 towers, to_qkv: q, k and v leave as the ATTENTION kernel's operand format - per aligned group of 8 features the 32 bytes
    // (hi0..7 | lo0..7), so that a 16-byte chunk is 8 consecutive hi (or lo) values: what its LDS-DMA copies and its fragment loads take as
    // they are.  A lane's 4 features are half a group: 8 bytes of hi, 8 bytes of lo.  The gate columns (between the rotary ranges) stay fp32.
    const int f_first = __builtin_amdgcn_readfirstlane(feat[0]);
    const bool plain = f_first >= p.rope_q_end && f_first < p.rope_k_begin;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (tv[j] && fv[i]) {
          if (plain) {
            *reinterpret_cast<f32x4*>((float*)p.y + (size_t)tok[j] * p.ldy + feat[i]) = acc[i][j];
          } else {
            const uint4 im = split4_bf16(__builtin_bit_cast(uint4, acc[i][j]));
            char* g8 = reinterpret_cast<char*>((float*)p.y + (size_t)tok[j] * p.ldy + (feat[i] & ~7)) + ((feat[i] >> 2) & 1) * 8;
            *reinterpret_cast<uint2*>(g8) = make_uint2(im.x, im.y);
            *reinterpret_cast<uint2*>(g8 + 16) = make_uint2(im.z, im.w);
          }
        }
  } else if (sizeof(T) == 4 && p.y_image) {
    // split-bf16 towers: a lane's four consecutive features are one 16-byte chunk of the NEXT linear's split image (hi0..3 | lo0..3)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (tv[j] && fv[i])
          *reinterpret_cast<uint4*>((float*)p.y + (size_t)tok[j] * p.ldy + feat[i]) = split4_bf16(__builtin_bit_cast(uint4, acc[i][j]));
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (tv[j] && fv[i]) Vec4<T>::store((T*)p.y + (size_t)tok[j] * p.ldy + feat[i], acc[i][j]);
  }
}

// ================================================================================================
// bf16 MFMA kernel, general K
// ================================================================================================
#define TF 128
#define TT 128
#define BK 64

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// NJ = 16-token fragments per wave: the token tile is 32 NJ rows (128 or 160).  The 160-row variant exists for grid balance:
// with 2 resident blocks per CU (512 slots) a 36 864-token, 256-feature GEMM is 576 tiles of 128 tokens = two rounds with the
// second one 12 % full, but 462 tiles of 160 tokens = one round (host picks the cheaper of the two).  The fifth staging chunk /
// fragment is written out with named scalars like the other four: with arrays hipcc keeps the staging registers in scratch and
// turns the global loads into flat loads (measured: +15 % on the whole training step).
template <int EPI, bool GATHER = false, int NJ = 4>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FT = DUAL ? 64 : TF;  // output features per block
  constexpr int TTK = 32 * NJ;        // tokens per block
  constexpr bool X5 = NJ == 5;
  static_assert(NJ == 4 || NJ == 5, "token tile is 128 or 160 rows");
  __shared__ uint4 lds[2][(TF + TTK) * 8];  // [buffer][w rows 0..127 | x rows][row*8 + swizzled chunk], 64 / 72 KiB
#define LDSW(buf_, idx_) lds[buf_][idx_]
#define LDSX(buf_, idx_) lds[buf_][TF * 8 + (idx_)]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wf = wave & 1, wt = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FT;
  const int tbase = (tile / n_ftiles) * TTK;

  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;

  // per-thread staging: 4 chunks (16 B) of each operand per k-tile; chunk = tid + 256*i -> row = chunk>>3, kc = chunk&7
  const int srow = tid >> 3, skc = tid & 7;
  const bf16_t* wp0; const bf16_t* wp1; const bf16_t* wp2; const bf16_t* wp3;
  const bf16_t* xp0; const bf16_t* xp1; const bf16_t* xp2; const bf16_t* xp3; const bf16_t* xp4 = nullptr;
  {
    auto wrow = [&](int row) {
      int wr;
      if (DUAL) wr = row < 64 ? fbase + row : p.N + fbase + (row - 64);
      else wr = fbase + row;
      wr = wr < p.w_rows ? wr : p.w_rows - 1;
      return W + (size_t)wr * p.ldw + skc * 8;
    };
    auto xrow = [&](int row) {
      int xr = tbase + row;
      xr = xr < p.M ? xr : p.M - 1;
      return X + (size_t)xr * p.ldx + skc * 8;
    };
    wp0 = wrow(srow); wp1 = wrow(srow + 32); wp2 = wrow(srow + 64); wp3 = wrow(srow + 96);
    xp0 = xrow(srow); xp1 = xrow(srow + 32); xp2 = xrow(srow + 64); xp3 = xrow(srow + 96);
    if (X5) xp4 = xrow(srow + 128);
  }
  // GATHER (encoder proj_in, blocks.py:91-93 + utils.py:26-34): GEMM row t is patch t and its K = (c, pt, ph, pw) vector is
  // read straight from the clip: a 16-byte chunk = the pw = 8 pixels of one (c, ipt, iph) image row of the patch
  PatchDst gx[NJ];
  if (GATHER) {
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      int t = tbase + srow + 32 * i;
      t = t < p.M ? t : p.M - 1;
      const int ci = p.row_seq[p.patch_rows[t]];
      const int* ds = p.clip_desc + (size_t)ci * 8;
      const int Tn = ds[0], H = ds[1], W = ds[2], gh = ds[4], gw = ds[5], pl = t - ds[6];
      const int gwi = pl % gw, r = pl / gw, ghi = r % gh, gti = r / gh;
      gx[i].thw = Tn * H * W; gx[i].hw = H * W; gx[i].w = W;
      gx[i].base = reinterpret_cast<bf16_t*>(p.clips.p[ci - p.clip0]) + ((size_t)((gti << p.pt_shift) * H + (ghi << p.ph_shift)) * W + gwi * 8);
    }
  }
#define GXPTR(i_, k0_)                                                                                          \
  ({                                                                                                            \
    const int sg__ = ((k0_) >> 3) + skc;                                                                        \
    const int c__ = sg__ >> (p.pt_shift + p.ph_shift), ipt__ = (sg__ >> p.ph_shift) & ((1 << p.pt_shift) - 1);  \
    const int iph__ = sg__ & ((1 << p.ph_shift) - 1);                                                           \
    (const bf16_t*)(gx[i_].base + c__ * gx[i_].thw + ipt__ * gx[i_].hw + iph__ * gx[i_].w);                      \
  })
  // swizzled LDS slot of (row, kc): row*8 + (kc ^ ((row>>1)&7))
  const int li0 = srow * 8 + (skc ^ ((srow >> 1) & 7));
  const int li1 = (srow + 32) * 8 + (skc ^ (((srow + 32) >> 1) & 7));
  const int li2 = (srow + 64) * 8 + (skc ^ (((srow + 64) >> 1) & 7));
  const int li3 = (srow + 96) * 8 + (skc ^ (((srow + 96) >> 1) & 7));
  const int li4 = (srow + 128) * 8 + (skc ^ (((srow + 128) >> 1) & 7));

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 sw0, sw1, sw2, sw3, sx0, sx1, sx2, sx3, sx4;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
#define GLOAD(k0)                                                                   \
  do {                                                                              \
    const bool ok__ = ((k0) + skc * 8) < p.K;                                       \
    sw0 = ok__ ? *reinterpret_cast<const uint4*>(wp0 + (k0)) : zero4;               \
    sw1 = ok__ ? *reinterpret_cast<const uint4*>(wp1 + (k0)) : zero4;               \
    sw2 = ok__ ? *reinterpret_cast<const uint4*>(wp2 + (k0)) : zero4;               \
    sw3 = ok__ ? *reinterpret_cast<const uint4*>(wp3 + (k0)) : zero4;               \
    sx0 = ok__ ? *reinterpret_cast<const uint4*>(GATHER ? GXPTR(0, k0) : xp0 + (k0)) : zero4; \
    sx1 = ok__ ? *reinterpret_cast<const uint4*>(GATHER ? GXPTR(1, k0) : xp1 + (k0)) : zero4; \
    sx2 = ok__ ? *reinterpret_cast<const uint4*>(GATHER ? GXPTR(2, k0) : xp2 + (k0)) : zero4; \
    sx3 = ok__ ? *reinterpret_cast<const uint4*>(GATHER ? GXPTR(3, k0) : xp3 + (k0)) : zero4; \
    if (X5) sx4 = ok__ ? *reinterpret_cast<const uint4*>(GATHER ? GXPTR(NJ - 1, k0) : xp4 + (k0)) : zero4; \
  } while (0)
#define LSTORE(buf)                                                                 \
  do {                                                                              \
    LDSW(buf, li0) = sw0; LDSW(buf, li1) = sw1; LDSW(buf, li2) = sw2; LDSW(buf, li3) = sw3; \
    LDSX(buf, li0) = sx0; LDSX(buf, li1) = sx1; LDSX(buf, li2) = sx2; LDSX(buf, li3) = sx3; \
    if (X5) LDSX(buf, li4) = sx4;                                                   \
  } while (0)

  const int l15 = lane & 15, kq = lane >> 4;
  const int nk = (p.K + BK - 1) / BK;
  GLOAD(0);
  LSTORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GLOAD((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[NJ];
      const int kc = ks * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
        a[i] = __builtin_bit_cast(bf16x8, LDSW(buf, arow * 8 + (kc ^ ((arow >> 1) & 7))));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int brow = wt * (16 * NJ) + j * 16 + l15;
        b[j] = __builtin_bit_cast(bf16x8, LDSX(buf, brow * 8 + (kc ^ ((brow >> 1) & 7))));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef GLOAD
#undef LSTORE
#undef GXPTR
#undef LDSW
#undef LDSX

  int tok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) tok[j] = tbase + wt * (16 * NJ) + j * 16 + l15;
  if (DUAL) {
    int feat[2];
    f32x4 ax[2][NJ], ag[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      feat[i] = fbase + wf * 32 + i * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { ax[i][j] = acc[i][j]; ag[i][j] = acc[i + 2][j]; }
    }
    epilogue_tile<EPI, bf16_t, 2, NJ>(p, tok, feat, ax, ag, kq);
  } else {
    int feat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) feat[i] = fbase + wf * 64 + i * 16 + kq * 4;
    epilogue_tile<EPI, bf16_t, 4, NJ>(p, tok, feat, acc, acc, kq);
  }
}

// ================================================================================================
// The same 128-feature x 128/160-token x 64-K structure with both operand tiles staged by LDS-DMA (global_load_lds_dwordx4) instead
// of through registers: no staging VGPRs, no ds_write pass, the next k-tile's loads are issued right after the barrier and land
// behind the whole MFMA phase.  One wave-instruction fills 1 KiB = 8 tile rows of 128 bytes, lane-linear in LDS, so the XOR
// swizzle of the fragment reads is applied on the SOURCE side (lane l of an instruction fetches chunk (l & 7) ^ ((row >> 1) & 7)
// of row l >> 3); the k-tile is a scalar base, the lane's share of it constant byte offsets (clamped rows included).
// Used when K % 64 == 0 and the operands fit 32-bit byte offsets (every linear of the base / large towers); the register-staged
// kernel above keeps the other cases (patch gather, K tails).
// ================================================================================================
template <int EPI, int NJ>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16_dma(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FT = DUAL ? 64 : TF;
  constexpr int TTK = 32 * NJ;
  constexpr int XI = TTK / 32;                  // X DMA instructions per wave and k-tile (8 rows each): 4 or 5
  __shared__ __attribute__((aligned(16))) uint4 lds[2][(TF + TTK) * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wave & 1, wt = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FT;
  const int tbase = (tile / n_ftiles) * TTK;
  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;

  // lane-constant source byte offsets: W instruction i of this wave covers tile rows 32 wave + 8 i + (lane >> 3)
  const int lr = lane >> 3, lp = lane & 7;
  uint32_t woff[4], xoff[XI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lr;
    int wr;
    if (DUAL) wr = row < 64 ? fbase + row : p.N + fbase + (row - 64);
    else wr = fbase + row;
    wr = wr < p.w_rows ? wr : p.w_rows - 1;
    woff[i] = ((uint32_t)wr * (uint32_t)p.ldw + (uint32_t)((lp ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = wave * (8 * XI) + i * 8 + lr;
    int xr = tbase + row;
    xr = xr < p.M ? xr : p.M - 1;
    xoff[i] = ((uint32_t)xr * (uint32_t)p.ldx + (uint32_t)((lp ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0][0];
  constexpr uint32_t BUFB = (TF + TTK) * 128;   // bytes per buffer
#define GD_DMA(voff_, base_, dst_)                                                                               \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
#define GD_STAGE(kt_, buf_)                                                                                      \
  do {                                                                                                           \
    const bf16_t* wb__ = W + (size_t)(kt_) * BK;                                                                 \
    const bf16_t* xb__ = X + (size_t)(kt_) * BK;                                                                 \
    const uint32_t dw__ = lds0 + (buf_) * BUFB + wave * 4096;                                                    \
    const uint32_t dx__ = lds0 + (buf_) * BUFB + TF * 128 + wave * (1024 * XI);                                  \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) GD_DMA(woff[i__], wb__, dw__ + i__ * 1024);              \
    _Pragma("unroll") for (int i__ = 0; i__ < XI; ++i__) GD_DMA(xoff[i__], xb__, dx__ + i__ * 1024);             \
  } while (0)

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, kq = lane >> 4;
  const int nk = p.K / BK;
  GD_STAGE(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GD_STAGE(kt + 1, buf ^ 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[NJ];
      const int kc = ks * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
        a[i] = __builtin_bit_cast(bf16x8, lds[buf][arow * 8 + (kc ^ ((arow >> 1) & 7))]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int brow = wt * (16 * NJ) + j * 16 + l15;
        b[j] = __builtin_bit_cast(bf16x8, lds[buf][TF * 8 + brow * 8 + (kc ^ ((brow >> 1) & 7))]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the next k-tile has landed
    __syncthreads();
  }
#undef GD_STAGE
#undef GD_DMA

  int tok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) tok[j] = tbase + wt * (16 * NJ) + j * 16 + l15;
  if (DUAL) {
    int feat[2];
    f32x4 ax[2][NJ], ag[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      feat[i] = fbase + wf * 32 + i * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { ax[i][j] = acc[i][j]; ag[i][j] = acc[i + 2][j]; }
    }
    epilogue_tile<EPI, bf16_t, 2, NJ>(p, tok, feat, ax, ag, kq);
  } else {
    int feat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) feat[i] = fbase + wf * 64 + i * 16 + kq * 4;
    epilogue_tile<EPI, bf16_t, 4, NJ>(p, tok, feat, acc, acc, kq);
  }
}

// ================================================================================================
// 256-feature x 256-token x 64-K tiles, 8 waves (2 feature halves x 4 token quarters; a wave owns 128 features x 64 tokens = 8 x 4
// accumulator tiles, 64 MFMAs per k-tile), both operand tiles staged by LDS-DMA into two 64 KiB stages: one block per CU.
// Against the 128 x 128 kernel above a fragment read from LDS feeds more MFMAs (24 ds_read_b128 per 64 MFMAs instead of 16 per 32),
// a wave issues 8 LDS-DMA instructions per 64 MFMAs instead of per 32, and a k-tile is long enough (64 MFMAs = 1 024 matrix cycles
// per wave) for the next tile's loads to land behind it.  cdna_hip_programming.md ("the 256^2 8-phase template") is the published
// recipe for this tile shape; the schedule here is this file's own (see the loop).
// EPI_GEGLU: a block covers 128 output features; tile rows [128 wr, 128 wr + 64) are x features fbase + 64 wr .., rows + 64 their gates.
// ================================================================================================
#define T256_F 256
#define T256_T 256
#define T256_DEFAULT 1
template <int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm_bf16_t256(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FOUT = DUAL ? 128 : T256_F;          // output features per block
  __shared__ __attribute__((aligned(16))) uint4 lds[2][(T256_F + T256_T) * 8];     // 2 x 64 KiB: [stage][row * 8 + swizzled 16-byte chunk]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FOUT;
  const int tbase = (tile / n_ftiles) * T256_T;
  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;

  // staging: the stage image is 512 rows of 128 bytes (256 weight rows, then 256 token rows); wave w stages rows 64 w .. 64 w + 63
  // with 8 instructions of 8 rows each (waves 0-3 the weights, 4-7 the tokens); swizzle on the source side as in k_gemm_bf16_dma
  const int lr = lane >> 3, lp = lane & 7;
  uint32_t soff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave & 3) * 64 + i * 8 + lr;      // row within the operand's 256
    const int ch = (lp ^ ((row >> 1) & 7)) * 8;
    if (wave < 4) {
      int wrow;
      if (DUAL) { const int h = row >> 7, rr = row & 127; wrow = rr < 64 ? fbase + h * 64 + rr : p.N + fbase + h * 64 + (rr - 64); }
      else wrow = fbase + row;
      wrow = wrow < p.w_rows ? wrow : p.w_rows - 1;
      soff[i] = ((uint32_t)wrow * (uint32_t)p.ldw + (uint32_t)ch) * 2u;
    } else {
      int xr = tbase + row;
      xr = xr < p.M ? xr : p.M - 1;
      soff[i] = ((uint32_t)xr * (uint32_t)p.ldx + (uint32_t)ch) * 2u;
    }
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0][0];
  constexpr uint32_t STAGEB = (T256_F + T256_T) * 128;
  const bf16_t* const sbase = wave < 4 ? W : X;
#define T2_STAGE(kt_, buf_)                                                                                      \
  do {                                                                                                           \
    const bf16_t* b__ = sbase + (size_t)(kt_) * BK;                                                              \
    const uint32_t d__ = lds0 + (buf_) * STAGEB + wave * 8192;                                                   \
    _Pragma("unroll") for (int i__ = 0; i__ < 8; ++i__) {                                                        \
      unsigned keep__;                                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep__) : "v"(soff[i__]), "s"(b__), "s"(d__ + i__ * 1024) : "memory");                 \
    }                                                                                                            \
  } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, kq = lane >> 4;
  const int nk = p.K / BK;

  // ---- schedule ----------------------------------------------------------------------------------------------------------
  // A k-tile is four PHASES, one per quadrant of the wave's 128 x 64 outputs: (A0,B0) (A0,B1) (A1,B1) (A1,B0), A0 / A1 = the wave's
  // feature tiles 0-3 / 4-7, B0 / B1 = its token tiles 0-1 / 2-3.  A phase is a short LOAD segment (this wave's LDS-DMA pieces, the
  // wait for the fragments the quadrant needs), a barrier, a MATRIX segment (16 MFMAs at raised priority, and between them the
  // ds_read_b128 of the fragments the NEXT phase adds - 4, 8, 0, 12 - into registers the running MFMAs do not use), a barrier.
  // The two wave groups (wr = 0 / 1: waves w and w + 4 share a SIMD) run ONE barrier apart - group 1 takes an extra barrier first,
  // group 0 an extra one at the end - so on every SIMD one wave is in its matrix segment while its partner is in its load segment.
  // Staging: tile t lives in stage t & 1.  A wave issues its 8 pieces of tile t + 2 in the load segments of phase 4 of tile t (6)
  // and phase 2 of tile t + 1 (2).
  //   WAR: the last reads of a stage's WEIGHT rows are phase 2's (A1, issued in its matrix segment and retired by the lgkmcnt(0) of
  //        phase 3's load segment), at least one barrier before a weight-staging wave (group 0) issues its first piece in phase 4.
  //        The last reads of its TOKEN rows are phase 3's re-read of B0.  Group 0's are retired by the lgkmcnt(0) of its phase-4 load
  //        segment, one barrier before the token-staging waves (group 1) reach theirs; group 1's OWN re-read is only one s_barrier
  //        ahead of its pieces, so the phase-4 load segment retires it explicitly (s_waitcnt lgkmcnt(0)) BEFORE the pieces are issued
  //        (until then this rested on in-order LDS issue and the DMA's latency).
  //   RAW: the first reads of tile t + 1 are group 0's, in the matrix segment of its phase 4.  Every wave has confirmed its own pieces
  //        of tile t + 1 (vmcnt(0)) before the barrier that opens that segment: group 0 at the top of its phase-4 load segment,
  //        group 1 - one barrier behind - at the end of its phase-3 matrix segment.  Both waits come before the wave issues its next
  //        pieces, so vmcnt(0) is exact, and the pieces waited for were issued 3 to 8 barrier intervals earlier.
#define T2_FENCE() __builtin_amdgcn_sched_barrier(0)
#define T2_BAR() do { T2_FENCE(); asm volatile("s_barrier" ::: "memory"); T2_FENCE(); } while (0)
#define T2_LOADDONE() do { T2_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); T2_FENCE(); } while (0)
#define T2_VMDONE() do { T2_FENCE(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T2_FENCE(); } while (0)
#define T2_LGKDONE() do { T2_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); T2_FENCE(); } while (0)
#define T2_PIECES(kt_, buf_, i0_, i1_)                                                                           \
  do {                                                                                                           \
    const bf16_t* b__ = sbase + (size_t)(kt_) * BK;                                                              \
    const uint32_t d__ = lds0 + (buf_) * STAGEB + wave * 8192;                                                   \
    _Pragma("unroll") for (int i__ = (i0_); i__ < (i1_); ++i__) {                                                \
      unsigned keep__;                                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep__) : "v"(soff[i__]), "s"(b__), "s"(d__ + i__ * 1024) : "memory");                 \
    }                                                                                                            \
  } while (0)
  // fragment addresses: row 16 i + l15 (+ a multiple of 64) of an operand, 16-byte chunk (4 ks + kq) ^ ((row >> 1) & 7) - and (row >> 1) & 7 is
  // (l15 >> 1) & 7 for every tile i, wave and operand, so two lane-constant byte offsets (ks = 0, 1) serve every read; the rest is an immediate
  const char* const ldsb = reinterpret_cast<const char*>(&lds[0][0]);
  const uint32_t foff0 = (uint32_t)(l15 * 128 + (((0 * 4 + kq) ^ ((l15 >> 1) & 7)) << 4)) + (uint32_t)(wr * 128 * 128);
  const uint32_t foff1 = (uint32_t)(l15 * 128 + (((1 * 4 + kq) ^ ((l15 >> 1) & 7)) << 4)) + (uint32_t)(wr * 128 * 128);
  const uint32_t goff0 = (uint32_t)(l15 * 128 + (((0 * 4 + kq) ^ ((l15 >> 1) & 7)) << 4)) + (uint32_t)(T256_F * 128 + wc * 64 * 128);
  const uint32_t goff1 = (uint32_t)(l15 * 128 + (((1 * 4 + kq) ^ ((l15 >> 1) & 7)) << 4)) + (uint32_t)(T256_F * 128 + wc * 64 * 128);
#define T2_READA(dst_, buf_, half_)                                                                              \
  do {                                                                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                        \
      dst_[0][i__] = *reinterpret_cast<const bf16x8*>(ldsb + foff0 + ((buf_) * STAGEB + ((half_) * 4 + i__) * 2048)); \
      dst_[1][i__] = *reinterpret_cast<const bf16x8*>(ldsb + foff1 + ((buf_) * STAGEB + ((half_) * 4 + i__) * 2048)); \
    }                                                                                                            \
  } while (0)
#define T2_READB(dst_, buf_, half_)                                                                              \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__) {                                                        \
      dst_[0][j__] = *reinterpret_cast<const bf16x8*>(ldsb + goff0 + ((buf_) * STAGEB + ((half_) * 2 + j__) * 2048)); \
      dst_[1][j__] = *reinterpret_cast<const bf16x8*>(ldsb + goff1 + ((buf_) * STAGEB + ((half_) * 2 + j__) * 2048)); \
    }                                                                                                            \
  } while (0)
#define T2_MFMA(fa_, ahalf_, fb_, bhalf_)                                                                        \
  do {                                                                                                           \
    _Pragma("unroll") for (int ks__ = 0; ks__ < 2; ++ks__)                                                       \
      _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__)                                                        \
        _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__)                                                      \
          acc[(ahalf_) * 4 + i__][(bhalf_) * 2 + j__] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
              fa_[ks__][i__], fb_[ks__][j__], acc[(ahalf_) * 4 + i__][(bhalf_) * 2 + j__], 0, 0, 0);              \
  } while (0)
  // one k-tile.  On entry FBA_ holds B0 and fa0[0] the first k-step of A0 (read during the previous tile's phase 4, or before the loop);
  // FBB_ is free.  On exit FBB_ holds the next tile's B0 and fa0[0] its A0, first k-step.  At most 80 fragment registers are live at once
  // (128 accumulators beside them): B0 is read twice per tile rather than kept across phases 2 and 3, and only half of the next A0 is
  // read ahead in phase 4 (its second k-step follows at the top of phase 1, behind the first eight MFMAs).
#define T2_READA_KS(dst_, buf_, half_, ks_)                                                                      \
  do {                                                                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__)                                                          \
      dst_[ks_][i__] = *reinterpret_cast<const bf16x8*>(ldsb + ((ks_) ? foff1 : foff0) + ((buf_) * STAGEB + ((half_) * 4 + i__) * 2048)); \
  } while (0)
#define T2_TILE(kt_, buf_, FBA_, FBB_)                                                                           \
  do {                                                                                                           \
    /* phase 1: (A0, B0); reads the second k-step of A0 and B1 */                                                \
    T2_LOADDONE();                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    T2_READA_KS(fa0, buf_, 0, 1);                                                                                \
    T2_READB(FBB_, buf_, 1);                                                                                     \
    T2_MFMA(fa0, 0, FBA_, 0);                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    T2_BAR();                                                                                                    \
    /* phase 2: (A0, B1); the last two pieces of tile kt + 1; reads A1 */                                        \
    if ((kt_) >= 1 && (kt_) + 1 < nk) T2_PIECES((kt_) + 1, (buf_) ^ 1, 6, 8);                                    \
    T2_LOADDONE();                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    T2_READA(fa1, buf_, 1);                                                                                      \
    T2_MFMA(fa0, 0, FBB_, 1);                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    T2_BAR();                                                                                                    \
    /* phase 3: (A1, B1); reads B0 again; group 1 confirms tile kt + 1 at the end */                             \
    T2_LOADDONE();                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    T2_READB(FBA_, buf_, 0);                                                                                     \
    T2_MFMA(fa1, 1, FBB_, 1);                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    if (wr == 1) T2_VMDONE();                                                                                    \
    T2_BAR();                                                                                                    \
    /* phase 4: (A1, B0); group 0 confirms tile kt + 1; six pieces of tile kt + 2 into this tile's stage; reads the next B0 and half of the next A0 */ \
    if (wr == 0) T2_VMDONE();                                                                                    \
    if ((kt_) + 2 < nk) { T2_LGKDONE(); T2_PIECES((kt_) + 2, buf_, 0, 6); }                                      \
    T2_LOADDONE();                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    if ((kt_) + 1 < nk) { T2_READB(FBB_, (buf_) ^ 1, 0); T2_READA_KS(fa0, (buf_) ^ 1, 0, 0); }                    \
    T2_MFMA(fa1, 1, FBA_, 0);                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    T2_BAR();                                                                                                    \
  } while (0)

  // prologue: tiles 0 and 1 staged whole
  T2_PIECES(0, 0, 0, 8);
  if (nk > 1) T2_PIECES(1, 1, 0, 8);
  T2_VMDONE();
  T2_BAR();
  if (wr == 1) T2_BAR();                // the stagger
  bf16x8 fa0[2][4], fa1[2][4], fbx[2][2], fby[2][2];
  T2_READB(fbx, 0, 0);
  T2_READA_KS(fa0, 0, 0, 0);
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {        // two tiles per trip: the two B register sets swap roles
    T2_TILE(kt, 0, fbx, fby);
    T2_TILE(kt + 1, 1, fby, fbx);
  }
  if (kt < nk) T2_TILE(kt, 0, fbx, fby);
  if (wr == 0) T2_BAR();                // balances group 1's extra barrier
#undef T2_FENCE
#undef T2_BAR
#undef T2_LOADDONE
#undef T2_VMDONE
#undef T2_LGKDONE
#undef T2_PIECES
#undef T2_READA
#undef T2_READB
#undef T2_READA_KS
#undef T2_MFMA
#undef T2_TILE
#undef T2_STAGE

  int tok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tok[j] = tbase + wc * 64 + j * 16 + l15;
  if (DUAL) {
    int feat[4];
    f32x4 ax[4][4], ag[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      feat[i] = fbase + wr * 64 + i * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { ax[i][j] = acc[i][j]; ag[i][j] = acc[i + 4][j]; }
    }
    epilogue_tile<EPI, bf16_t, 4, 4>(p, tok, feat, ax, ag, kq);
  } else if (EPI == EPI_QKV_ROPE) {
    // a wave's 128 features are two heads with the SAME rotary factors (they depend on the token and on the feature's position
    // inside its head): loaded once - the 8-byte loads of the [L,64] table are what this epilogue costs - and applied to both
    // halves, then stored by the plain epilogue.  Arithmetic and order as in epilogue_tile (row scale first, then the rotation).
    const int f_first = __builtin_amdgcn_readfirstlane(fbase + wr * 128);
    if (!(p.debug & 1) && (f_first < p.rope_q_end || (f_first >= p.rope_k_begin && f_first < p.rope_k_end))) {
      float2 c[4][4], sn[4][4];
      float rs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tc = tok[j] < p.M ? tok[j] : p.M - 1;
        rs[j] = p.row_scale ? p.row_scale[tc] : 1.0f;
        const float* cs = p.rope_cs + (size_t)tc * 64 + kq * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c[i][j] = *reinterpret_cast<const float2*>(cs + i * 8);
          sn[i][j] = *reinterpret_cast<const float2*>(cs + 32 + i * 8);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 a = acc[hf * 4 + i][j];
            if (p.row_scale) a *= rs[j];
            acc[hf * 4 + i][j] = (f32x4){a[0] * c[i][j].x - a[1] * sn[i][j].x, a[0] * sn[i][j].x + a[1] * c[i][j].x,
                                         a[2] * c[i][j].y - a[3] * sn[i][j].y, a[2] * sn[i][j].y + a[3] * c[i][j].y};
          }
      GemmDev q = p;
      q.row_scale = nullptr;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int feat[4];
        f32x4 ah[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          feat[i] = fbase + wr * 128 + hf * 64 + i * 16 + kq * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) ah[i][j] = acc[hf * 4 + i][j];
        }
        epilogue_tile<EPI_STORE, bf16_t, 4, 4>(q, tok, feat, ah, ah, kq);     // bias / add_scalar are null for to_qkv
      }
    } else {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int feat[4];
        f32x4 ah[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          feat[i] = fbase + wr * 128 + hf * 64 + i * 16 + kq * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) ah[i][j] = acc[hf * 4 + i][j];
        }
        epilogue_tile<EPI_STORE, bf16_t, 4, 4>(p, tok, feat, ah, ah, kq);
      }
    }
  } else {
    // two halves of 64 features: half the epilogue's side registers (residual rows) live at a time
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      int feat[4];
      f32x4 ah[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        feat[i] = fbase + wr * 128 + hf * 64 + i * 16 + kq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) ah[i][j] = acc[hf * 4 + i][j];
      }
      epilogue_tile<EPI, bf16_t, 4, 4>(p, tok, feat, ah, ah, kq);
    }
  }
}

// ================================================================================================
// Mixed bf16 / fp8 linears (BASELINE config #5): both operands in OCP e4m3 with one fp32 scale per row (token / weight row;
// k_quant_rows_fp8), products on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales - twice the bf16 MFMA rate and half the
// operand bytes through L2 and LDS - fp32 accumulation, the two row scales applied to the accumulator before the usual epilogue
// (output bf16).  A 128-element k-tile of fp8 is 128 bytes per row: the LDS image, its XOR swizzle and the LDS-DMA staging are
// byte for byte those of k_gemm_bf16_dma.  K order inside a k-tile: lane group kq supplies 16-byte chunks kq and kq + 4 of its
// row as its 32 operand bytes - any assignment works as long as both operands use the same one (the instruction pairs element j
// of lane group kq of A with element j of lane group kq of B).
// ================================================================================================
typedef int v8i32 __attribute__((ext_vector_type(8)));

// MX = true (round 4): both operands carry one E8M0 scale per 32 CONSECUTIVE k elements (k_quant_mx_fp8), fed to the instruction's
// per-lane scale operands.  Measured on the part (tools/ubench/mx_probe.hip, profiles/r04_mx_probe.txt): the instruction's k order is
// "lane group g holds bytes 16 g .. 16 g + 15 and 64 + 16 g .. 64 + 16 g + 15 of the 128-byte row" - exactly the chunk pair (kq, kq + 4)
// this kernel reads - and the scale byte of lane (row, group b) multiplies k-block b = bytes 32 b .. 32 b + 31 of that row (the low
// halves of groups 2b', 2b'+1 for b < 2, the high halves for b >= 2), NOT the 32 bytes the lane itself holds; op_sel picks the byte.
// So the operand image stays as it is and lane (row, kq) supplies the scale of block 4 kt + kq: a lane's scale bytes of four
// consecutive k-tiles arrive as one dword per operand row (prefetched one group ahead) and are shifted into byte 0.  The per-row
// fp32 factors become optional (activations: the rstd of the folded pre-norm, p.x_scale; weights: the row factor of the pack).
template <int EPI, int NJ, bool MX = false>
__global__ __launch_bounds__(256, 2) void k_gemm_fp8_dma(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FT = DUAL ? 64 : TF;
  constexpr int TTK = 32 * NJ;
  constexpr int XI = TTK / 32;
  constexpr int BK8 = 128;                      // fp8 elements = bytes per k-tile and row
  __shared__ __attribute__((aligned(16))) uint4 lds[2][(TF + TTK) * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wave & 1, wt = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FT;
  const int tbase = (tile / n_ftiles) * TTK;
  const uint8_t* W = (const uint8_t*)p.w;
  const uint8_t* X = (const uint8_t*)p.x;

  const int lr = lane >> 3, lp = lane & 7;
  uint32_t woff[4], xoff[XI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lr;
    int wr;
    if (DUAL) wr = row < 64 ? fbase + row : p.N + fbase + (row - 64);
    else wr = fbase + row;
    wr = wr < p.w_rows ? wr : p.w_rows - 1;
    woff[i] = (uint32_t)wr * (uint32_t)p.ldw + (uint32_t)((lp ^ ((row >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = wave * (8 * XI) + i * 8 + lr;
    int xr = tbase + row;
    xr = xr < p.M ? xr : p.M - 1;
    xoff[i] = (uint32_t)xr * (uint32_t)p.ldx + (uint32_t)((lp ^ ((row >> 1) & 7)) * 16);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0][0];
  constexpr uint32_t BUFB = (TF + TTK) * 128;
#define G8_DMA(voff_, base_, dst_)                                                                               \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
#define G8_STAGE(kt_, buf_)                                                                                      \
  do {                                                                                                           \
    const uint8_t* wb__ = W + (size_t)(kt_) * BK8;                                                               \
    const uint8_t* xb__ = X + (size_t)(kt_) * BK8;                                                               \
    const uint32_t dw__ = lds0 + (buf_) * BUFB + wave * 4096;                                                    \
    const uint32_t dx__ = lds0 + (buf_) * BUFB + TF * 128 + wave * (1024 * XI);                                  \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) G8_DMA(woff[i__], wb__, dw__ + i__ * 1024);              \
    _Pragma("unroll") for (int i__ = 0; i__ < XI; ++i__) G8_DMA(xoff[i__], xb__, dx__ + i__ * 1024);             \
  } while (0)

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, kq = lane >> 4;
  const int nk = p.K / BK8;
  // MX: scale dwords (4 k-tiles each) of this lane's operand rows; `cur` serves the running group of four k-tiles, `nxt` the next
  uint32_t sa_cur[4], sb_cur[NJ], sa_nxt[4], sb_nxt[NJ];
  const uint8_t* sa_ptr[4];
  const uint8_t* sb_ptr[NJ];
  if constexpr (MX) {
    const int nkp = p.ld_mx >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int wrow = (DUAL ? (i < 2 ? fbase + wf * 32 + i * 16 : p.N + fbase + wf * 32 + (i - 2) * 16) : fbase + wf * 64 + i * 16) + l15;
      wrow = wrow < p.w_rows ? wrow : p.w_rows - 1;
      sa_ptr[i] = p.w_mx + (size_t)wrow * p.ld_mx + kq * nkp;
      sa_cur[i] = *reinterpret_cast<const uint32_t*>(sa_ptr[i]);
      sa_nxt[i] = sa_cur[i];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int xr = tbase + wt * (16 * NJ) + j * 16 + l15;
      xr = xr < p.M ? xr : p.M - 1;
      sb_ptr[j] = p.x_mx + (size_t)xr * p.ld_mx + kq * nkp;
      sb_cur[j] = *reinterpret_cast<const uint32_t*>(sb_ptr[j]);
      sb_nxt[j] = sb_cur[j];
    }
  }
  G8_STAGE(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) G8_STAGE(kt + 1, buf ^ 1);
    if constexpr (MX) {
      if ((kt & 3) == 0) {
        if (kt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sa_cur[i] = sa_nxt[i];
#pragma unroll
          for (int j = 0; j < NJ; ++j) sb_cur[j] = sb_nxt[j];
        }
        if (kt + 4 < nk) {                          // the next group's bytes: in flight for four k-tiles
#pragma unroll
          for (int i = 0; i < 4; ++i) sa_nxt[i] = *reinterpret_cast<const uint32_t*>(sa_ptr[i] + kt + 4);
#pragma unroll
          for (int j = 0; j < NJ; ++j) sb_nxt[j] = *reinterpret_cast<const uint32_t*>(sb_ptr[j] + kt + 4);
        }
      }
    }
    v8i32 a[4], b[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
      const int sw = (arow >> 1) & 7;
      const uint4 lo = lds[buf][arow * 8 + (kq ^ sw)], hi = lds[buf][arow * 8 + ((kq + 4) ^ sw)];
      a[i] = (v8i32){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int brow = wt * (16 * NJ) + j * 16 + l15;
      const int sw = (brow >> 1) & 7;
      const uint4 lo = lds[buf][TF * 8 + brow * 8 + (kq ^ sw)], hi = lds[buf][TF * 8 + brow * 8 + ((kq + 4) ^ sw)];
      b[j] = (v8i32){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
    if constexpr (MX) {
      const int sh = (kt & 3) * 8;
      int sa[4], sb[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) sa[i] = (int)(sa_cur[i] >> sh);
#pragma unroll
      for (int j = 0; j < NJ; ++j) sb[j] = (int)(sb_cur[j] >> sh);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0 /* byte 0 */, sa[i], 0, sb[j]);
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0 /* A: fp8 e4m3 */, 0 /* B: fp8 e4m3 */,
                                                                      0, 0x7F7F7F7F /* E8M0 1.0 */, 0, 0x7F7F7F7F);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
#undef G8_STAGE
#undef G8_DMA

  // dequantise: acc *= x_scale[token] * w_scale[feature]
  int tok[NJ];
  float sx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    tok[j] = tbase + wt * (16 * NJ) + j * 16 + l15;
    sx[j] = (MX && !p.x_scale) ? 1.0f : p.x_scale[tok[j] < p.M ? tok[j] : p.M - 1];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int f;
    if (DUAL) f = (i < 2 ? fbase + wf * 32 + i * 16 : p.N + fbase + wf * 32 + (i - 2) * 16) + kq * 4;
    else f = fbase + wf * 64 + i * 16 + kq * 4;
    f = f + 3 < p.w_rows ? f : p.w_rows - 4;
    const f32x4 sw4 = (MX && !p.w_scale) ? (f32x4){1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(p.w_scale + f);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] *= sx[j] * sw4[e];
  }
  if (DUAL) {
    if (MX && p.yq) {
      // GEGLU output straight into the NEXT linear's operand format (round 4): a wave's two 16-feature tiles of a token are exactly one
      // 32-element MX block of h - spread over the four lane groups kq -, so the block maximum is two xor shuffles away, and the
      // bf16 copy of h (151 MB at the base shape) and its quantisation pass disappear.  Values are rounded to bf16 first: the image is
      // bit for bit what k_quant_mx_fp8 makes of the bf16 h the plain epilogue stores.
      if (p.debug & 1) return;
      const int blk = (fbase + wf * 32) >> 5;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h0[e] = round_to<bf16_t>(geglu_fast(acc[2][j][e], acc[0][j][e]));
          h1[e] = round_to<bf16_t>(geglu_fast(acc[3][j][e], acc[1][j][e]));
        }
        float a = fmaxf(fmaxf(fmaxf(fabsf(h0[0]), fabsf(h0[1])), fmaxf(fabsf(h0[2]), fabsf(h0[3]))),
                        fmaxf(fmaxf(fabsf(h1[0]), fabsf(h1[1])), fmaxf(fabsf(h1[2]), fabsf(h1[3]))));
        a = quad16_max(a);
        const uint32_t tb = __float_as_uint(a * (1.0f / 448.0f));
        int byte = (int)((tb >> 23) & 0xFF) + ((tb & 0x7FFFFF) ? 1 : 0);
        byte = a > 0.f ? (byte < 1 ? 1 : (byte > 254 ? 254 : byte)) : 127;
        const float inv = __uint_as_float((uint32_t)(254 - byte) << 23);
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(h0[0] * inv, h0[1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(h0[2] * inv, h0[3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(h1[0] * inv, h1[1] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(h1[2] * inv, h1[3] * inv, w1, true);
        const int t = tok[j];
        if (t < p.M && fbase + wf * 32 + 32 <= p.N) {
          uint8_t* row = p.yq + (size_t)t * p.N + fbase + wf * 32 + kq * 4;
          *reinterpret_cast<int*>(row) = w0;
          *reinterpret_cast<int*>(row + 16) = w1;
          if (kq == 0) p.yq_mx[(size_t)t * p.ld_yq_mx + (blk & 3) * p.yq_nkp + (blk >> 2)] = (uint8_t)byte;
        }
      }
      return;
    }
    int feat[2];
    f32x4 ax[2][NJ], ag[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      feat[i] = fbase + wf * 32 + i * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { ax[i][j] = acc[i][j]; ag[i][j] = acc[i + 2][j]; }
    }
    epilogue_tile<EPI, bf16_t, 2, NJ>(p, tok, feat, ax, ag, kq);
  } else {
    int feat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) feat[i] = fbase + wf * 64 + i * 16 + kq * 4;
    epilogue_tile<EPI, bf16_t, 4, NJ>(p, tok, feat, acc, acc, kq);
  }
}

// ================================================================================================
// bf16 MFMA kernel for K == 256
//
// 4 waves, each owning 32 tokens of a 128-token tile with the whole K in registers (2 n-tiles x 8 k-steps x 16 B =
// 64 VGPRs) and computing them against ALL 64 rows of the current weight panel (4 m-tiles): per item 64 MFMAs per
// wave, A fragments re-read from the shared LDS panel, B fragments never re-read.  A wave therefore writes
// 64 (32 for GEGLU) consecutive output features per token.  Item order per block: contiguous (tile, panel) range.
// Loop body: prefetch next panel (global->regs) | 64 MFMA | stage next panel to LDS | epilogue | barrier - the
// panel's vmcnt wait sits BEFORE the epilogue's stores so it never drains them (vmcnt retires in issue order).
//
// PRENORM: the preceding RMSNorm (transformer.py:86 / :48) is folded in: its gain is pre-multiplied into the weight
// columns on the host (w' = w * gain), and rstd = rsqrt(mean(x^2)+eps) is computed here from the register-resident
// token row (in-lane sum + two xor shuffles) and applied to the fp32 accumulator - x_normed is never materialised.
// ================================================================================================
#define K256_TT 128
#define K256_ROWS 64

template <int EPI, bool PRENORM>
__global__ __launch_bounds__(256, 2) void k_gemm_k256(GemmDev p, int n_panels, int total_items) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FO = DUAL ? 32 : 64;  // output features per panel
  __shared__ uint4 wl[2][K256_ROWS * 32];  // [buffer][row*32 + swizzled 16-byte chunk], 2 x 32 KiB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations are SGPR operands
  const int l15 = lane & 15, kq = lane >> 4;
  // consecutive logical blocks share token tiles; keep them on one XCD (same L2) under round-robin dispatch
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int it0 = (int)((long)total_items * lb / gridDim.x);
  const int it1 = (int)((long)total_items * (lb + 1) / gridDim.x);
  if (it0 >= it1) return;

  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;

  // panel staging by LDS-DMA (global_load_lds_dwordx4): one wave-instruction fills 1 KiB = two 512-byte panel rows,
  // lane-linear in LDS, so the XOR swizzle the fragment reads use is applied to the SOURCE chunk instead
  // (same involution on both sides).  Wave w stages rows 16w .. 16w+15 (8 instructions), no VGPR staging, no ds_write.
  const int g_half = lane >> 5, g_c = lane & 31;
#define WROWIDX(panel_, row_)                                                                                \
  ({                                                                                                         \
    int wr__ = DUAL ? ((row_) < 32 ? (panel_) * 32 + (row_) : p.N + (panel_) * 32 + ((row_) - 32)) : (panel_) * 64 + (row_); \
    wr__ < p.w_rows ? wr__ : p.w_rows - 1;                                                                   \
  })
  // issued as asm (scalar base + per-lane byte offset): with the builtin form hipcc sees an LDS write in flight and turns every
  // lgkmcnt wait of the main loop into lgkmcnt(0), exposing a fragment-read latency every other k-step
  const uint32_t wl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&wl[0][0];
#define GLDS_PANEL(panel_, buf_)                                                                             \
  do {                                                                                                       \
    _Pragma("unroll") for (int i__ = 0; i__ < 8; ++i__) {                                                    \
      const int r0__ = wave * 16 + 2 * i__, row__ = r0__ + g_half;                                           \
      const int ch__ = (g_c & 16) | ((g_c & 15) ^ (row__ & 15));                                             \
      const uint32_t voff__ = ((uint32_t)WROWIDX(panel_, row__) * (uint32_t)p.ldw + ch__ * 8) * 2u;          \
      const uint32_t dst__ = wl_lds + ((buf_) * (K256_ROWS * 32) + r0__ * 32) * 16;                          \
      unsigned keep__;                                                                                       \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep__) : "v"(voff__), "s"(p.w), "s"(dst__) : "memory");                             \
    }                                                                                                        \
  } while (0)
  // A fragments of k-step s8 (4 m-tiles) from panel buffer buf_
#define LOADA(dst_, buf_, s8_)                                                                               \
  do {                                                                                                       \
    const int ch__ = (s8_) * 4 + kq;                                                                         \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                    \
      const int arow__ = i__ * 16 + l15;                                                                     \
      dst_[i__] = __builtin_bit_cast(bf16x8, wl[buf_][arow__ * 32 + ((ch__ & 16) | ((ch__ & 15) ^ (arow__ & 15)))]); \
    }                                                                                                        \
  } while (0)
#define MFMA8(a_, s8_)                                                                                       \
  do {                                                                                                       \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__)                                                      \
      _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__)                                                    \
        acc[i__][j__] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[i__], bfr[j__][s8_], acc[i__][j__], 0, 0, 0); \
  } while (0)

  bf16x8 bfr[2][8];
  float rstd[2] = {1.f, 1.f};
  int cur_tile = -1;
  // EPI_QKV_ROPE: every rotary panel is one 64-wide head, so the (cos, sin) pairs this lane needs depend only on the
  // token: feature i*16 + kq*4 + {0..3} -> complex pairs i*8 + kq*2 + {0,1}.  Kept in registers per token tile.
  float2 rc[4][2], rs[4][2];
  PatchDst pdst[2];

#ifndef K256_DEPTH
#define K256_DEPTH 2      // A-fragment register buffers: fragments of k-step s + DEPTH - 1 are requested before the MFMAs of step s.
                          // -DK256_DEPTH=3 (two steps ahead, +12..17 VGPRs, no scratch) measured neutral in a same-box A/B, round 4
                          // (QKV 35.9 / 36.0 vs 36.2 / 36.4 us on a slow box): the fragment-read latency is not what the loop waits for
#endif
  bf16x8 af[K256_DEPTH][4];
  GLDS_PANEL(it0 % n_panels, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  LOADA(af[0], 0, 0);
  if (K256_DEPTH == 3) LOADA(af[1], 0, 1);
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) & 1;
    const int tile = it / n_panels, panel = it - tile * n_panels;
    if (tile != cur_tile && !((p.debug & 4) && cur_tile >= 0)) {
      cur_tile = tile;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int t = tile * K256_TT + wave * 32 + j * 16 + l15;
        t = t < p.M ? t : p.M - 1;
        if (p.x_rows) t = p.x_rows[t];
        const bf16_t* xr = X + (size_t)t * p.ldx + kq * 8;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) bfr[j][s8] = *reinterpret_cast<const bf16x8*>(xr + s8 * 32);
      }
      if (EPI == EPI_QKV_ROPE) {
        if (p.rope_ids) {
          // factors by position id: the row's three ids (8 bytes) instead of its 256-byte fp32 row; the (cos, sin) pairs come from the
          // base table (n_ids x 10 pairs, L2 / L1 resident).  Complex pair pp = 8 i + 2 kq + e of a head rotates by frequency pp / 3
          // of axis pp % 3 (rope.py:40-54: column f * 3 + axis); pairs 30, 31 take the identity row (id slot 3).
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            int t = tile * K256_TT + wave * 32 + j * 16 + l15;
            t = t < p.M ? t : p.M - 1;
            const uint2 idp = *reinterpret_cast<const uint2*>(p.rope_ids + 2 * (size_t)t);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float2 cs2[2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int pp = i * 8 + kq * 2 + e;
                const int f = pp < 30 ? pp / 3 : 0, axis = pp < 30 ? pp - 3 * (pp / 3) : 3;
                // v_perm_b32: bytes (2 axis, 2 axis + 1) of {idp.y : idp.x} into the low half, zero above
                const uint32_t id = __builtin_amdgcn_perm(idp.y, idp.x, 0x0c0c0000u | (uint32_t)((2 * axis + 1) << 8) | (uint32_t)(2 * axis));
                cs2[e] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.rope_base) + id * 80u + (uint32_t)f * 8u);
              }
              rc[i][j] = make_float2(cs2[0].x, cs2[1].x);
              rs[i][j] = make_float2(cs2[0].y, cs2[1].y);
            }
          }
        } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int t = tile * K256_TT + wave * 32 + j * 16 + l15;
          t = t < p.M ? t : p.M - 1;
          const float* cs = p.rope_cs + (size_t)t * 64 + kq * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            rc[i][j] = *reinterpret_cast<const float2*>(cs + i * 8);
            rs[i][j] = *reinterpret_cast<const float2*>(cs + 32 + i * 8);
          }
        }
        }
      }
      if (EPI == EPI_STORE_PATCH) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int t = tile * K256_TT + wave * 32 + j * 16 + l15;
          t = t < p.M ? t : p.M - 1;
          const int ci = p.row_seq[p.patch_rows[t]];
          const int* ds = p.clip_desc + (size_t)ci * 8;
          const int Tn = ds[0], H = ds[1], W = ds[2], gh = ds[4], gw = ds[5], pl = t - ds[6];
          const int gwi = pl % gw, r = pl / gw, ghi = r % gh, gti = r / gh;
          pdst[j].thw = Tn * H * W; pdst[j].hw = H * W; pdst[j].w = W;
          pdst[j].base = reinterpret_cast<bf16_t*>(p.clips.p[ci - p.clip0]) +
                         ((size_t)((gti << p.pt_shift) * H + (ghi << p.ph_shift)) * W + gwi * 8);
        }
      }
      if (PRENORM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float ss = 0.f;
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = (float)bfr[j][s8][e];
              ss = fmaf(v, v, ss);
            }
          ss = quad16_sum(ss);
          rstd[j] = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
        }
      }
    }
    // next panel -> other buffer, in flight behind the whole MFMA phase (every wave left that buffer before the
    // previous item's barrier)
    if (it + 1 < it1 && !(p.debug & 2)) GLDS_PANEL((it + 1) % n_panels, buf ^ 1);

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A fragments double-buffered in registers: the reads of k-step s+1 are issued before the 8 MFMAs of k-step s
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      constexpr int AH = K256_DEPTH - 1;     // how many k-steps ahead the fragments are requested
      if (s8 + AH < 8) {
        const int ch = (s8 + AH) * 4 + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int arow = i * 16 + l15;
          af[(s8 + AH) % K256_DEPTH][i] = __builtin_bit_cast(bf16x8, wl[buf][arow * 32 + ((ch & 16) | ((ch & 15) ^ (arow & 15)))]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s8 % K256_DEPTH][i], bfr[j][s8], acc[i][j], 0, 0, 0);
      if (s8 + AH < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ONE barrier per item, before the epilogue: next panel landed (vmcnt(0)) and every wave is done with this one;
    // the waves then run their epilogues unsynchronised, with the first fragments of the next item already requested
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the next panel is in LDS
    __syncthreads();
    if (it + 1 < it1) {
      LOADA(af[0], buf ^ 1, 0);
      if (K256_DEPTH == 3) LOADA(af[1], buf ^ 1, 1);
    }

    int tok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      tok[j] = tile * K256_TT + wave * 32 + j * 16 + l15;
      if (PRENORM && !(p.debug & 8)) {     // debug bit 8 (tools/qkv_ablate.py): no epilogue arithmetic - timing only, garbage results
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] *= rstd[j];
      }
    }
    if (DUAL) {  // panel rows 0..31 = x features, 32..63 = their gate rows
      int feat[2] = {panel * FO + kq * 4, panel * FO + 16 + kq * 4};
      f32x4 ax[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
      f32x4 ag[2][2] = {{acc[2][0], acc[2][1]}, {acc[3][0], acc[3][1]}};
      epilogue_tile<EPI, bf16_t, 2, 2>(p, tok, feat, ax, ag, kq);
    } else if (EPI == EPI_QKV_ROPE) {
      int feat[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) feat[i] = panel * FO + i * 16 + kq * 4;
      const int f_first = panel * FO;
      if (!(p.debug & 8) && (f_first < p.rope_q_end || (f_first >= p.rope_k_begin && f_first < p.rope_k_end))) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 a = acc[i][j];
            acc[i][j] = (f32x4){a[0] * rc[i][j].x - a[1] * rs[i][j].x, a[0] * rs[i][j].x + a[1] * rc[i][j].x,
                                a[2] * rc[i][j].y - a[3] * rs[i][j].y, a[2] * rs[i][j].y + a[3] * rc[i][j].y};
          }
      }
      epilogue_tile<EPI_STORE, bf16_t, 4, 2>(p, tok, feat, acc, acc, kq);   // bias / add_scalar are null for to_qkv
    } else {
      int feat[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) feat[i] = panel * FO + i * 16 + kq * 4;
      epilogue_tile<EPI, bf16_t, 4, 2>(p, tok, feat, acc, acc, kq, EPI == EPI_STORE_PATCH ? pdst : nullptr);
    }
  }
#undef WROWIDX
#undef GLDS_PANEL
#undef LOADA
#undef MFMA8
}

#include "ttv_qkv256.inc"
#include "ttv_qkv256ws.inc"

// ================================================================================================
// out_proj of a width-256 tower fused with the whole KEEL step (transformer.py:141-142):
//     x_new = RMSNorm(alpha * x + ao @ Wo^T) * gain          (K == 256, N == 256, bf16)
// Same register-resident-token structure as k_gemm_k256, but a wave keeps the accumulators of ALL 256 output features of
// its 16 tokens (16 m-tiles), so the row statistics are wave-local: in-lane sum of 64 squares + two xor shuffles.
// Nothing is written in fp32 and no separate norm pass exists.  The 4 weight panels cycle 0,1,2,3,0,1,... across tiles,
// so the LDS double-buffer prefetch never drains at a tile boundary.
// ================================================================================================
#define ROW_TT 64

__global__ __launch_bounds__(256, 2) void k_gemm_k256_rownorm(GemmDev p, int n_tiles) {
  __shared__ uint4 wl[2][K256_ROWS * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  if ((int)blockIdx.x >= n_tiles) return;
  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;
  const int srow = tid >> 5, sch = tid & 31;
  uint4 st0, st1, st2, st3, st4, st5, st6, st7;
#define WROWN(panel_, i_) (*reinterpret_cast<const uint4*>(W + (size_t)((panel_) * 64 + srow + 8 * (i_)) * p.ldw + sch * 8))
#define GLOADN(panel_)                                                                                       \
  do {                                                                                                       \
    st0 = WROWN(panel_, 0); st1 = WROWN(panel_, 1); st2 = WROWN(panel_, 2); st3 = WROWN(panel_, 3);          \
    st4 = WROWN(panel_, 4); st5 = WROWN(panel_, 5); st6 = WROWN(panel_, 6); st7 = WROWN(panel_, 7);          \
  } while (0)
#define LIDXN(i_) ((srow + 8 * (i_)) * 32 + ((sch & 16) | ((sch & 15) ^ ((srow + 8 * (i_)) & 15))))
#define LSTOREN(buf_)                                                                                        \
  do {                                                                                                       \
    wl[buf_][LIDXN(0)] = st0; wl[buf_][LIDXN(1)] = st1; wl[buf_][LIDXN(2)] = st2; wl[buf_][LIDXN(3)] = st3;  \
    wl[buf_][LIDXN(4)] = st4; wl[buf_][LIDXN(5)] = st5; wl[buf_][LIDXN(6)] = st6; wl[buf_][LIDXN(7)] = st7;  \
  } while (0)

  GLOADN(0);
  LSTOREN(0);
  __syncthreads();
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const bool last_tile = tile + (int)gridDim.x >= n_tiles;
    int t = tile * ROW_TT + wave * 16 + l15;
    const bool tvalid = t < p.M;
    t = tvalid ? t : p.M - 1;
    bf16x8 bfr[8];
    {
      const bf16_t* xr = X + (size_t)t * p.ldx + kq * 8;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) bfr[s8] = *reinterpret_cast<const bf16x8*>(xr + s8 * 32);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int panel = 0; panel < 4; ++panel) {
      const int buf = panel & 1;   // 4 panels per tile: the buffer parity is the same for every tile
      const bool more = !(last_tile && panel == 3);
      if (more) GLOADN((panel + 1) & 3);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int ch = s8 * 4 + kq;
        bf16x8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int arow = i * 16 + l15;
          a[i] = __builtin_bit_cast(bf16x8, wl[buf][arow * 32 + ((ch & 16) | ((ch & 15) ^ (arow & 15)))]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[panel * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bfr[s8], acc[panel * 4 + i], 0, 0, 0);
      }
      if (more) LSTOREN(buf ^ 1);
      __syncthreads();
    }
    // ---- y = alpha*resid + acc ; x_new = y * rsqrt(mean(y^2)+eps) * gain ----
    const bf16_t* rrow = (const bf16_t*)p.resid + (size_t)t * p.ldr + kq * 4;
    f32x4 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = Vec4<bf16_t>::load(rrow + i * 16);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] += p.alpha * r[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) ss = fmaf(acc[i][e], acc[i][e], ss);
    }
    ss = quad16_sum(ss);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
    const bool odd = kq & 1;
    bf16_t* yrow = (bf16_t*)p.y + (size_t)t * p.ldy;
    if (p.sum_f32 && tvalid) {          // training tape: the pre-norm sum (uniform branch)
      float* srow_ = p.sum_f32 + (size_t)t * p.ld_sum + kq * 4;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(srow_ + i * 16) = acc[i];
    }
    float ss2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_gain + i * 16 + kq * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[i][e] = round_to<bf16_t>(acc[i][e] * rstd * g[e]);     // the value that is stored
        ss2 = fmaf(acc[i][e], acc[i][e], ss2);
      }
    }
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
      const uint2 p0 = pack_bf16x4(acc[i0]), p1 = pack_bf16x4(acc[i1]);
      const uint4 out = xchg16_pair(p0, p1);     // v_permlane16_swap: even kq (own p0, partner's p0), odd kq (partner's p1, own p1)
      const int start = odd ? i1 * 16 + kq * 4 - 4 : i0 * 16 + kq * 4;
      if (tvalid && !(p.debug & 1)) *reinterpret_cast<uint4*>(yrow + start) = out;
    }
    if (p.y2) {                         // the NEXT pre-norm of the row just written (training tape: xn of the following sub-layer)
      ss2 = quad16_sum(ss2);
      const float rstd2 = 1.0f / sqrtf(ss2 * (1.0f / 256.0f) + p.eps);
      bf16_t* y2row = (bf16_t*)p.y2 + (size_t)t * p.ldy2;
#pragma unroll
      for (int ip = 0; ip < 8; ++ip) {
        const int i0 = 2 * ip, i1 = 2 * ip + 1;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.norm_gain2 + i0 * 16 + kq * 4);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.norm_gain2 + i1 * 16 + kq * 4);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = acc[i0][e] * rstd2 * g0[e]; y1[e] = acc[i1][e] * rstd2 * g1[e]; }
        const uint2 p0 = pack_bf16x4(y0), p1 = pack_bf16x4(y1);
        const uint4 out = xchg16_pair(p0, p1);     // v_permlane16_swap: even kq (own p0, partner's p0), odd kq (partner's p1, own p1)
        const int start = odd ? i1 * 16 + kq * 4 - 4 : i0 * 16 + kq * 4;
        if (tvalid && !(p.debug & 1)) *reinterpret_cast<uint4*>(y2row + start) = out;
      }
    }
  }
#undef WROWN
#undef GLOADN
#undef LIDXN
#undef LSTOREN
}

// ================================================================================================
// Full-row tile kernel, general K, N == 256: y = RMSNorm(alpha*resid + x w^T) * gain   (w3 of a width-256 tower + the
// KEEL step, transformer.py:55,144-145).  Block = 256 features x 64 tokens, 4 waves each 64 features x 64 tokens
// (4x4 MFMA tiles), K-tiles of 64 through double-buffered swizzled LDS (W 32 KiB + X 8 KiB per buffer, 2 blocks / CU).
// Row statistics: in-lane sum over the lane's 16 features per token, xor-16/32 shuffles (wave = 64 features), then a
// 4-wave reduction through LDS.  Output bf16, 16-byte paired stores; nothing is written in fp32.
// ================================================================================================
__global__ __launch_bounds__(256, 2) void k_gemm_rowtile_norm(GemmDev p) {
  __shared__ uint4 lw[2][256 * 8];   // 2 x 32 KiB
  __shared__ uint4 lx[2][64 * 8];    // 2 x 8 KiB
  float (*red)[64] = reinterpret_cast<float (*)[64]>(&lx[0][0]);   // [4][64] row partial sums; aliases lx after the K loop (80 KiB total -> 2 blocks / CU)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int tbase = blockIdx.x * 64;
  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;
  // staging: W 2048 chunks (8 per thread: row = (tid>>3) + 32*i, chunk = tid&7), X 512 chunks (2 per thread)
  const int srow = tid >> 3, skc = tid & 7;
  int xr0 = tbase + srow, xr1 = tbase + srow + 32;
  xr0 = xr0 < p.M ? xr0 : p.M - 1;
  xr1 = xr1 < p.M ? xr1 : p.M - 1;
  const bf16_t* xp0 = X + (size_t)xr0 * p.ldx + skc * 8;
  const bf16_t* xp1 = X + (size_t)xr1 * p.ldx + skc * 8;
  const bf16_t* wp = W + (size_t)srow * p.ldw + skc * 8;
  uint4 sw0, sw1, sw2, sw3, sw4, sw5, sw6, sw7, sx0, sx1;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
#define RT_GLOAD(k0)                                                                              \
  do {                                                                                            \
    const bool ok__ = ((k0) + skc * 8) < p.K;                                                     \
    const bf16_t* w__ = wp + (k0);                                                                \
    const size_t st__ = (size_t)32 * p.ldw;                                                       \
    sw0 = ok__ ? *reinterpret_cast<const uint4*>(w__) : zero4;                                    \
    sw1 = ok__ ? *reinterpret_cast<const uint4*>(w__ + st__) : zero4;                             \
    sw2 = ok__ ? *reinterpret_cast<const uint4*>(w__ + 2 * st__) : zero4;                         \
    sw3 = ok__ ? *reinterpret_cast<const uint4*>(w__ + 3 * st__) : zero4;                         \
    sw4 = ok__ ? *reinterpret_cast<const uint4*>(w__ + 4 * st__) : zero4;                         \
    sw5 = ok__ ? *reinterpret_cast<const uint4*>(w__ + 5 * st__) : zero4;                         \
    sw6 = ok__ ? *reinterpret_cast<const uint4*>(w__ + 6 * st__) : zero4;                         \
    sw7 = ok__ ? *reinterpret_cast<const uint4*>(w__ + 7 * st__) : zero4;                         \
    sx0 = ok__ ? *reinterpret_cast<const uint4*>(xp0 + (k0)) : zero4;                             \
    sx1 = ok__ ? *reinterpret_cast<const uint4*>(xp1 + (k0)) : zero4;                             \
  } while (0)
#define RT_IDX(r_) ((r_) * 8 + (skc ^ (((r_) >> 1) & 7)))
#define RT_LSTORE(buf)                                                                            \
  do {                                                                                            \
    lw[buf][RT_IDX(srow)] = sw0; lw[buf][RT_IDX(srow + 32)] = sw1; lw[buf][RT_IDX(srow + 64)] = sw2; lw[buf][RT_IDX(srow + 96)] = sw3;       \
    lw[buf][RT_IDX(srow + 128)] = sw4; lw[buf][RT_IDX(srow + 160)] = sw5; lw[buf][RT_IDX(srow + 192)] = sw6; lw[buf][RT_IDX(srow + 224)] = sw7; \
    lx[buf][RT_IDX(srow)] = sx0; lx[buf][RT_IDX(srow + 32)] = sx1;                                \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = (p.K + BK - 1) / BK;
  RT_GLOAD(0);
  RT_LSTORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) RT_GLOAD((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
      const int kc = ks * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = wave * 64 + i * 16 + l15, brow = i * 16 + l15;
        a[i] = __builtin_bit_cast(bf16x8, lw[buf][arow * 8 + (kc ^ ((arow >> 1) & 7))]);
        b[i] = __builtin_bit_cast(bf16x8, lx[buf][brow * 8 + (kc ^ ((brow >> 1) & 7))]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) RT_LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef RT_GLOAD
#undef RT_IDX
#undef RT_LSTORE
  // ---- epilogue: y = alpha*resid + acc; row sum of squares over all 256 features; scale; gain; store ----
  float ssq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int t = tbase + j * 16 + l15;
    t = t < p.M ? t : p.M - 1;
    const bf16_t* rrow = (const bf16_t*)p.resid + (size_t)t * p.ldr + wave * 64 + kq * 4;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 r = Vec4<bf16_t>::load(rrow + i * 16);
      acc[i][j] += p.alpha * r;
#pragma unroll
      for (int e = 0; e < 4; ++e) ss = fmaf(acc[i][j][e], acc[i][j][e], ss);
    }
    ss = quad16_sum(ss);
    ssq[j] = ss;
  }
  if (kq == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j * 16 + l15] = ssq[j];
  }
  __syncthreads();
  const bool odd = kq & 1;
  float ssq2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tl = j * 16 + l15, t = tbase + tl;
    const float tot = red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl];
    const float rstd = 1.0f / sqrtf(tot * (1.0f / 256.0f) + p.eps);
    if (p.sum_f32 && t < p.M) {         // training tape: the pre-norm sum (uniform branch)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(p.sum_f32 + (size_t)t * p.ld_sum + wave * 64 + i * 16 + kq * 4) = acc[i][j];
    }
    float ss2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_gain + wave * 64 + i * 16 + kq * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[i][j][e] = round_to<bf16_t>(acc[i][j][e] * rstd * g[e]);   // the value that is stored
        ss2 = fmaf(acc[i][j][e], acc[i][j][e], ss2);
      }
    }
    ss2 = quad16_sum(ss2);
    ssq2[j] = ss2;
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
      const int f0 = wave * 64 + i0 * 16 + kq * 4, f1 = wave * 64 + i1 * 16 + kq * 4;
      const uint2 p0 = pack_bf16x4(acc[i0][j]), p1 = pack_bf16x4(acc[i1][j]);
      const uint4 out = xchg16_pair(p0, p1);     // v_permlane16_swap: even kq (own p0, partner's p0), odd kq (partner's p1, own p1)
      const int start = odd ? f1 - 4 : f0;
      if (t < p.M && !(p.debug & 1)) *reinterpret_cast<uint4*>((bf16_t*)p.y + (size_t)t * p.ldy + start) = out;
    }
  }
  if (p.y2) {                           // the NEXT pre-norm of the rows just written: a second exchange of the four waves' partial sums
    __syncthreads();                    // every wave has read `red`
    if (kq == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][j * 16 + l15] = ssq2[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tl = j * 16 + l15, t = tbase + tl;
      const float tot2 = red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl];
      const float rstd2 = 1.0f / sqrtf(tot2 * (1.0f / 256.0f) + p.eps);
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        const int i0 = 2 * ip, i1 = 2 * ip + 1;
        const int f0 = wave * 64 + i0 * 16 + kq * 4, f1 = wave * 64 + i1 * 16 + kq * 4;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.norm_gain2 + f0), g1 = *reinterpret_cast<const f32x4*>(p.norm_gain2 + f1);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = acc[i0][j][e] * rstd2 * g0[e]; y1[e] = acc[i1][j][e] * rstd2 * g1[e]; }
        const uint2 p0 = pack_bf16x4(y0), p1 = pack_bf16x4(y1);
        const uint4 out = xchg16_pair(p0, p1);     // v_permlane16_swap: even kq (own p0, partner's p0), odd kq (partner's p1, own p1)
        const int start = odd ? f1 - 4 : f0;
        if (t < p.M && !(p.debug & 1)) *reinterpret_cast<uint4*>((bf16_t*)p.y2 + (size_t)t * p.ldy2 + start) = out;
      }
    }
  }
}

// ================================================================================================
// fp32 kernel: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, one rounding per product = an fmaf chain;
// 64 FLOP/clk/SIMD = the fp32 vector peak, 157 TFLOP/s on MI355X).  This is the compute path of `dtype=float32` towers, the
// mode in which token indices equal the reference's fp32 result (fsq.py:123-135 rounds after tanh: bf16 cannot be bit-exact).
// Same orientation and C layout as the bf16 kernel (a lane owns 4 consecutive output features of one token), so the epilogues
// are shared.  128 features x 128 tokens x 32 K per step, 4 waves (2 x 2), register-staged double-buffered LDS; rows are 128
// bytes = 8 chunks of 16 bytes, chunk kc of row r stored at kc ^ (r & 7) (conflict-free ds_read_b128 for the lane -> (row, kq)
// fragment pattern, contiguous ds_write_b128).  A lane's 16-byte chunk holds k = 4 kq .. 4 kq + 3 of a 16-k block: MFMA step jj
// of the block therefore sums k in {jj, 4 + jj, 8 + jj, 12 + jj} - a fixed permutation of the summation order, the same
// for both operands.
// ================================================================================================
#define F_TF 128
#define F_TT 128
#define F_BK 32

// SPLIT (round 4; "index-exact at a third of the bf16 rate"): the same tiles, staging and epilogues, but every product runs as THREE
// bf16 MFMA passes on split operands, x = hi + lo with hi = bf16(x), lo = bf16(x - hi):  a b ~ ah bh + ah bl + al bh  (fp32 accumulation;
// the dropped al bl and the split's remainder are ~2^-17 relative - measured on the encoder: max |pre-rounding FSQ value error| 5.8e-4
// against the reference's fp32 run, every token index equal, tests/probes/split_bf16_probe.py).  The weight operand arrives pre-split
// (pack time): its fp32-sized image holds, per 16-byte chunk, (hi0..3 | lo0..3) of the chunk's four k values, so it is staged by the
// very same copies; the token operand is split by the staging threads between the global load and the LDS store (once per element and
// block).  A lane's chunks kq and kq + 4 of the 32-wide k-tile are the two halves of its 8-element bf16 fragment (hi: dwords x, y of
// both chunks, lo: z, w) - the same k assignment for both operands, so v_mfma_f32_16x16x32_bf16 pairs equal k.

template <int EPI, bool SPLIT = false, int NJ = 4>
__global__ __launch_bounds__(256, 2) void k_gemm_f32(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FT = DUAL ? 64 : F_TF;
  constexpr int TTK = 32 * NJ;                  // tokens per block: 128, or 160 where that saves a round of resident blocks (launch())
  __shared__ uint4 lds[2][(F_TF + TTK) * 8];    // [buffer][w rows 0..127 | x rows 0..TTK-1][row*8 + swizzled chunk]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wf = wave & 1, wt = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FT;
  const int tbase = (tile / n_ftiles) * TTK;
  const float* W = (const float*)p.w;
  const float* X = (const float*)p.x;

  // staging: 8 chunks per row and k-step; thread -> row (tid >> 3) + 32 i, chunk kc = tid & 7: 4 weight rows and NJ token rows per thread
  const int srow = tid >> 3, skc = tid & 7;
  // element offsets, not pointers: arrays of 64-bit pointers end up in scratch here (and every staging load then waits for a scratch
  // reload); 32-bit offsets stay in registers.  M * ldx and w_rows * ldw < 2^31 elements is checked by the launcher.
  uint32_t wo[4], xo[NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 32 * i;
    int wr;
    if (DUAL) wr = row < 64 ? fbase + row : p.N + fbase + (row - 64);
    else wr = fbase + row;
    wr = wr < p.w_rows ? wr : p.w_rows - 1;
    wo[i] = (uint32_t)wr * (uint32_t)p.ldw + (uint32_t)(skc * 4);
  }
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    int xr = tbase + srow + 32 * i;
    xr = xr < p.M ? xr : p.M - 1;
    xo[i] = (uint32_t)xr * (uint32_t)p.ldx + (uint32_t)(skc * 4);
  }
  const int li0 = srow * 8 + (skc ^ (srow & 7));            // (srow + 32 i) & 7 == srow & 7

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // named staging registers (sx4 only with the 160-token tile): written as arrays the compiler keeps some of them in scratch
  uint4 sw0, sw1, sw2, sw3, sx0, sx1, sx2, sx3, sx4;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
  const uint32_t wo0 = wo[0], wo1 = wo[1], wo2 = wo[2], wo3 = wo[3];
  const uint32_t xo0 = xo[0], xo1 = xo[1], xo2 = xo[2], xo3 = xo[3], xo4 = xo[NJ - 1];
#define FGLOAD(k0)                                                                  \
  do {                                                                              \
    const bool ok__ = ((k0) + skc * 4) < p.K;                                       \
    sw0 = ok__ ? *reinterpret_cast<const uint4*>(W + wo0 + (k0)) : zero4;           \
    sw1 = ok__ ? *reinterpret_cast<const uint4*>(W + wo1 + (k0)) : zero4;           \
    sw2 = ok__ ? *reinterpret_cast<const uint4*>(W + wo2 + (k0)) : zero4;           \
    sw3 = ok__ ? *reinterpret_cast<const uint4*>(W + wo3 + (k0)) : zero4;           \
    sx0 = ok__ ? *reinterpret_cast<const uint4*>(X + xo0 + (k0)) : zero4;           \
    sx1 = ok__ ? *reinterpret_cast<const uint4*>(X + xo1 + (k0)) : zero4;           \
    sx2 = ok__ ? *reinterpret_cast<const uint4*>(X + xo2 + (k0)) : zero4;           \
    sx3 = ok__ ? *reinterpret_cast<const uint4*>(X + xo3 + (k0)) : zero4;           \
    if (NJ == 5) sx4 = ok__ ? *reinterpret_cast<const uint4*>(X + xo4 + (k0)) : zero4; \
  } while (0)
#define FSPLIT(v_) ((SPLIT && !p.x_image) ? split4_bf16(v_) : (v_))
#define FLSTORE(buf)                                                                \
  do {                                                                              \
    lds[buf][li0] = sw0; lds[buf][li0 + 32 * 8] = sw1; lds[buf][li0 + 64 * 8] = sw2; lds[buf][li0 + 96 * 8] = sw3;            \
    lds[buf][F_TF * 8 + li0] = FSPLIT(sx0); lds[buf][F_TF * 8 + li0 + 32 * 8] = FSPLIT(sx1);                                  \
    lds[buf][F_TF * 8 + li0 + 64 * 8] = FSPLIT(sx2); lds[buf][F_TF * 8 + li0 + 96 * 8] = FSPLIT(sx3);                          \
    if (NJ == 5) lds[buf][F_TF * 8 + li0 + 128 * 8] = FSPLIT(sx4);                                                            \
  } while (0)

  const int l15 = lane & 15, kq = lane >> 4;
  const int nk = (p.K + F_BK - 1) / F_BK;
  FGLOAD(0);
  FLSTORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) FGLOAD((kt + 1) * F_BK);
    if constexpr (SPLIT) {
      bf16x8 ah[4], al[4], bh[NJ], bl[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
        const uint4 c0 = lds[buf][arow * 8 + (kq ^ (arow & 7))], c1 = lds[buf][arow * 8 + ((kq + 4) ^ (arow & 7))];
        ah[i] = __builtin_bit_cast(bf16x8, make_uint4(c0.x, c0.y, c1.x, c1.y));
        al[i] = __builtin_bit_cast(bf16x8, make_uint4(c0.z, c0.w, c1.z, c1.w));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int brow = wt * (16 * NJ) + j * 16 + l15;
        const uint4 c0 = lds[buf][F_TF * 8 + brow * 8 + (kq ^ (brow & 7))], c1 = lds[buf][F_TF * 8 + brow * 8 + ((kq + 4) ^ (brow & 7))];
        bh[j] = __builtin_bit_cast(bf16x8, make_uint4(c0.x, c0.y, c1.x, c1.y));
        bl[j] = __builtin_bit_cast(bf16x8, make_uint4(c0.z, c0.w, c1.z, c1.w));
      }
      // the two cross terms first, the leading term last (small + small + large)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x4 a[4], b[NJ];
        const int kc = kb * 4 + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
          a[i] = __builtin_bit_cast(f32x4, lds[buf][arow * 8 + (kc ^ (arow & 7))]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int brow = wt * (16 * NJ) + j * 16 + l15;
          b[j] = __builtin_bit_cast(f32x4, lds[buf][F_TF * 8 + brow * 8 + (kc ^ (brow & 7))]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][jj], b[j][jj], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) FLSTORE(buf ^ 1);
    __syncthreads();
  }
#undef FGLOAD
#undef FLSTORE
#undef FSPLIT

  int tok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) tok[j] = tbase + wt * (16 * NJ) + j * 16 + l15;
  if (DUAL) {
    int feat[2];
    f32x4 ax[2][NJ], ag[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      feat[i] = fbase + wf * 32 + i * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { ax[i][j] = acc[i][j]; ag[i][j] = acc[i + 2][j]; }
    }
    epilogue_tile<EPI, float, 2, NJ>(p, tok, feat, ax, ag, kq);
  } else {
    int feat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) feat[i] = fbase + wf * 64 + i * 16 + kq * 4;
    epilogue_tile<EPI, float, 4, NJ>(p, tok, feat, acc, acc, kq);
  }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 linear with BOTH operands staged by LDS-DMA (round 4): when the token operand already is a split image (its producer wrote
// it: GemmArgs.x_image) the k-tile of 32 k values is 128 bytes per row for both operands - byte for byte the stage image, swizzle and
// LDS-DMA staging of k_gemm_fp8_dma / k_gemm_bf16_dma: no staging registers, no ds_write, nothing to convert.  Fragments and the three
// MFMA passes as in k_gemm_f32<.., SPLIT>; fp32 epilogues.  K % 32 == 0.
// ------------------------------------------------------------------------------------------------
template <int EPI, int NJ, bool SPLIT = true>
__global__ __launch_bounds__(256, 2) void k_gemm_split_dma(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FT = DUAL ? 64 : TF;
  constexpr int TTK = 32 * NJ;
  constexpr int XI = TTK / 32;
  __shared__ __attribute__((aligned(16))) uint4 lds[2][(TF + TTK) * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wave & 1, wt = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FT;
  const int tbase = (tile / n_ftiles) * TTK;
  const char* W = (const char*)p.w;
  const char* X = (const char*)p.x;

  const int lr = lane >> 3, lp = lane & 7;
  uint32_t woff[4], xoff[XI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lr;
    int wr;
    if (DUAL) wr = row < 64 ? fbase + row : p.N + fbase + (row - 64);
    else wr = fbase + row;
    wr = wr < p.w_rows ? wr : p.w_rows - 1;
    woff[i] = ((uint32_t)wr * (uint32_t)p.ldw) * 4u + (uint32_t)((lp ^ ((row >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = wave * (8 * XI) + i * 8 + lr;
    int xr = tbase + row;
    xr = xr < p.M ? xr : p.M - 1;
    xoff[i] = ((uint32_t)xr * (uint32_t)p.ldx) * 4u + (uint32_t)((lp ^ ((row >> 1) & 7)) * 16);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0][0];
  constexpr uint32_t BUFB = (TF + TTK) * 128;
#define GS_DMA(voff_, base_, dst_)                                                                               \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
#define GS_STAGE(kt_, buf_)                                                                                      \
  do {                                                                                                           \
    const char* wb__ = W + (size_t)(kt_) * 128;                                                                  \
    const char* xb__ = X + (size_t)(kt_) * 128;                                                                  \
    const uint32_t dw__ = lds0 + (buf_) * BUFB + wave * 4096;                                                    \
    const uint32_t dx__ = lds0 + (buf_) * BUFB + TF * 128 + wave * (1024 * XI);                                  \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) GS_DMA(woff[i__], wb__, dw__ + i__ * 1024);              \
    _Pragma("unroll") for (int i__ = 0; i__ < XI; ++i__) GS_DMA(xoff[i__], xb__, dx__ + i__ * 1024);             \
  } while (0)

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, kq = lane >> 4;
  const int nk = p.K / 32;
  GS_STAGE(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GS_STAGE(kt + 1, buf ^ 1);
    if constexpr (SPLIT) {
    bf16x8 ah[4], al[4], bh[NJ], bl[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
      const int sw = (arow >> 1) & 7;
      const uint4 c0 = lds[buf][arow * 8 + (kq ^ sw)], c1 = lds[buf][arow * 8 + ((kq + 4) ^ sw)];
      ah[i] = __builtin_bit_cast(bf16x8, make_uint4(c0.x, c0.y, c1.x, c1.y));
      al[i] = __builtin_bit_cast(bf16x8, make_uint4(c0.z, c0.w, c1.z, c1.w));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int brow = wt * (16 * NJ) + j * 16 + l15;
      const int sw = (brow >> 1) & 7;
      const uint4 c0 = lds[buf][TF * 8 + brow * 8 + (kq ^ sw)], c1 = lds[buf][TF * 8 + brow * 8 + ((kq + 4) ^ sw)];
      bh[j] = __builtin_bit_cast(bf16x8, make_uint4(c0.x, c0.y, c1.x, c1.y));
      bl[j] = __builtin_bit_cast(bf16x8, make_uint4(c0.z, c0.w, c1.z, c1.w));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    } else {
      // exact fp32 (SPLIT = false): the same stage image holds plain fp32 rows; products and summation order of k_gemm_f32 (a lane's
      // chunk kc = 4 kb + kq holds k = 4 kc .. 4 kc + 3, MFMA step jj of a 16-k block sums k in {jj, 4 + jj, 8 + jj, 12 + jj})
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x4 a[4], b[NJ];
        const int kc = kb * 4 + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
          a[i] = __builtin_bit_cast(f32x4, lds[buf][arow * 8 + (kc ^ ((arow >> 1) & 7))]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int brow = wt * (16 * NJ) + j * 16 + l15;
          b[j] = __builtin_bit_cast(f32x4, lds[buf][TF * 8 + brow * 8 + (kc ^ ((brow >> 1) & 7))]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][jj], b[j][jj], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
#undef GS_STAGE
#undef GS_DMA

  int tok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) tok[j] = tbase + wt * (16 * NJ) + j * 16 + l15;
  if (DUAL) {
    int feat[2];
    f32x4 ax[2][NJ], ag[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      feat[i] = fbase + wf * 32 + i * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { ax[i][j] = acc[i][j]; ag[i][j] = acc[i + 2][j]; }
    }
    epilogue_tile<EPI, float, 2, NJ>(p, tok, feat, ax, ag, kq);
  } else {
    int feat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) feat[i] = fbase + wf * 64 + i * 16 + kq * 4;
    epilogue_tile<EPI, float, 4, NJ>(p, tok, feat, acc, acc, kq);
  }
}

template <int EPI>
static int launch(const GemmDev& d, int dtype, bool prenorm, hipStream_t s) {
  // debug bit 16384 (diagnostics): send a K = 256 GEMM through the general-K kernels instead (no folded pre-norm, no patch scatter there)
  if (dtype == TTV_BF16 && d.K == 256 && d.N % 8 == 0 && !((d.debug & 16384) && !prenorm && EPI != EPI_STORE_PATCH)) {
    const int fo = (EPI == EPI_GEGLU) ? 32 : 64;
    const int n_panels = ttv_cdiv(d.N, fo), n_tiles = ttv_cdiv(d.M, K256_TT);
    const int total = n_panels * n_tiles;
    // 2 co-resident blocks per CU = 512 slots.  Blocks take contiguous (tile, panel) item ranges and reload their token rows
    // whenever a range crosses into the next tile (-5 us of 34.6 with that reload knocked out).  With an item count per block that
    // is an integer, the ranges repeat with a short period against the panels of a tile and fewer of them straddle: QKV at the
    // benchmark batch, 3456 items = 288 tiles x 12 panels: 512 blocks x 6.75 items - more than half of the blocks straddle -
    // 34.6 us; 432 blocks x 8 items - one in three - 33.6 us (576 x 6, aligned but more than 512 slots: 35.2; 384 x 9: 37.4).
    // So: the largest grid in [416, 512] that divides the item count, else 512 (a grid much below the slot count loses more
    // than it gains: GEGLU, 6336 items, 396 x 16: 46 us against 41).  TTV_K256_GRID overrides (A/B).
    static const int grid_env = getenv("TTV_K256_GRID") ? atoi(getenv("TTV_K256_GRID")) : 0;
    int grid = total < 512 ? total : 512;
    if (grid_env > 0) grid = total < grid_env ? total : grid_env;
    else if (total > 512)
      for (int g = 512; g >= 416; --g)
        if (total % g == 0) { grid = g; break; }
    // to_qkv: the wave-pipelined streaming kernel k_qkv256 (ttv_qkv256.inc) unless the call needs something only the general epilogue
    // has (30.3 us against k_gemm_k256's 31.2 inside the benchmark forward, profiles/r04_qkv256_inpipe.txt).  TTV_QKV256=0 / ttv_debug_set
    // bit 15 (32768): k_gemm_k256's QKV instantiation; TTV_QKV256=2 / bit 17 (131072): the weight-stationary kernel k_qkv256ws
    // (ttv_qkv256ws.inc; 34.2 us: measured, kept for A/B) - all three give the same bits without the folded pre-norm
    static const int qkv256_env = getenv("TTV_QKV256") ? atoi(getenv("TTV_QKV256")) : 1;
    if constexpr (EPI == EPI_QKV_ROPE) {
      const int n_groups = (n_panels + 3) / 4;
      if (qkv256_env && !(d.debug & 32768) && d.N % 64 == 0 && d.rope_q_end % 64 == 0 && d.rope_k_begin % 64 == 0 && d.rope_k_end % 64 == 0 &&
          !d.row_scale && !d.bias && !d.add_scalar && (uint64_t)d.w_rows * (uint64_t)d.ldw * 2u < (1ull << 32)) {
        if ((qkv256_env >= 2 || (d.debug & 131072)) && n_groups <= 32) {
          // one block per CU: blocks b, b + 8, .. (one XCD under round-robin dispatch) share an eighth of the 32-token groups, split between
          // the panel groups; no more blocks than there are units for
          const int tgs = ttv_cdiv(d.M, 32), per_xcd = ttv_cdiv(tgs, 8);
          int q = n_groups * ttv_cdiv(per_xcd, 8);
          q = q > 32 ? 32 : q;
          if (prenorm) hipLaunchKernelGGL((k_qkv256ws<true>), dim3(8 * q), dim3(512), 0, s, d, n_panels, n_groups);
          else hipLaunchKernelGGL((k_qkv256ws<false>), dim3(8 * q), dim3(512), 0, s, d, n_panels, n_groups);
          TTV_CHECK_LAUNCH("qkv256ws");
          return TTV_OK;
        }
        if (prenorm) hipLaunchKernelGGL((k_qkv256<true>), dim3(grid), dim3(256), 0, s, d, n_panels, total);
        else hipLaunchKernelGGL((k_qkv256<false>), dim3(grid), dim3(256), 0, s, d, n_panels, total);
        TTV_CHECK_LAUNCH("qkv256");
        return TTV_OK;
      }
    }
    if (prenorm) hipLaunchKernelGGL((k_gemm_k256<EPI, true>), dim3(grid), dim3(256), 0, s, d, n_panels, total);
    else hipLaunchKernelGGL((k_gemm_k256<EPI, false>), dim3(grid), dim3(256), 0, s, d, n_panels, total);
    TTV_CHECK_LAUNCH("gemm_k256");
    return TTV_OK;
  }
  if (prenorm || EPI == EPI_STORE_PATCH) {
    ttv_set_error("gemm: folded pre-norm / patch scatter need the bf16 K=256 kernel");
    return TTV_ERR_UNSUPPORTED;
  }
  if (dtype == TTV_BF16) {
    const int ft = (EPI == EPI_GEGLU) ? 64 : TF;
    const int nf = ttv_cdiv(d.N, ft), nt = ttv_cdiv(d.M, TT), nt160 = ttv_cdiv(d.M, 160);
    // rounds of 512 resident blocks x tile height: the 160-token tile when it saves a (mostly empty) round
    const int force_tt = (d.debug & 128) ? 160 : (d.debug & 256) ? 128 : 0;   // diagnostics / tests (ttv_debug_set)
    const long cost128 = (long)ttv_cdiv(nf * nt, 512) * 128, cost160 = (long)ttv_cdiv(nf * nt160, 512) * 160;
    // both tiles by LDS-DMA when the k range is whole 64-element tiles and a lane's byte offset fits 32 bits (TTV_GEMM_DMA=0: A/B)
    static const bool use_dma = !(getenv("TTV_GEMM_DMA") && getenv("TTV_GEMM_DMA")[0] == '0');
    const bool dma_ok = use_dma && d.K % BK == 0 && (uint64_t)d.M * (uint64_t)d.ldx * 2u < (1ull << 32) && (uint64_t)d.w_rows * (uint64_t)d.ldw * 2u < (1ull << 32);
    // large GEMMs of the wide towers: 256 x 256 tiles (k_gemm_bf16_t256) when the grid gives every CU at least one tile.  Measured at
    // the base tower's shapes (36 864 rows, tools/gemm_t256_bench.py): w12 + GEGLU 294 -> 239 us, w3 + residual 154 -> 138, out_proj +
    // residual 76 -> 74, to_qkv + rotary 190 -> 187, plain store 121 -> 121; bench.py --config base 97.6 -> 102.6 clips/s.  With the
    // epilogue's stores knocked out both kernels run their k loops at 0.42-0.50 of the bf16 peak: what separates the cases is the
    // epilogue (fp32 residual sums and rotary factors cost 35-80 us per launch).  TTV_GEMM_T256=0|1 forces (A/B)
    static const int t256_env = getenv("TTV_GEMM_T256") ? atoi(getenv("TTV_GEMM_T256")) : -1;
    if constexpr (EPI == EPI_STORE || EPI == EPI_QKV_ROPE || EPI == EPI_GEGLU || EPI == EPI_RESID_T || EPI == EPI_RESID_F32) {
      const int fo = (EPI == EPI_GEGLU) ? 128 : T256_F;
      const long tiles256 = (long)ttv_cdiv(d.N, fo) * ttv_cdiv(d.M, T256_T);
      // ttv_debug_set bit 512 forces the 256 x 256 kernel wherever it is applicable (tests, A/B), bit 1024 forbids it
      const bool forced = (d.debug & 512) || t256_env == 1;
      const bool ok256 = dma_ok && d.N % fo == 0 && d.K >= 2 * BK && !(d.debug & 1024) && (forced || tiles256 >= 256);
      if (ok256 && (forced || (t256_env < 0 && T256_DEFAULT))) {
        hipLaunchKernelGGL((k_gemm_bf16_t256<EPI>), dim3((unsigned)tiles256), dim3(512), 0, s, d, ttv_cdiv(d.N, fo));
        TTV_CHECK_LAUNCH("gemm_t256");
        return TTV_OK;
      }
    }
    if ((cost160 < cost128 && force_tt != 128) || force_tt == 160) {
      if (dma_ok) hipLaunchKernelGGL((k_gemm_bf16_dma<EPI, 5>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
      else hipLaunchKernelGGL((k_gemm_bf16<EPI, false, 5>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
    } else {
      if (dma_ok) hipLaunchKernelGGL((k_gemm_bf16_dma<EPI, 4>), dim3(nf * nt), dim3(256), 0, s, d, nf);
      else hipLaunchKernelGGL((k_gemm_bf16<EPI>), dim3(nf * nt), dim3(256), 0, s, d, nf);
    }
  } else {
    // 128- or 160-token tiles by the same rounds-of-512-resident-blocks rule as the bf16 kernel: N = 256 at the benchmark batch is 576
    // tiles of 128 tokens - two rounds, the second an eighth full - and 462 tiles of 160 in one
    if ((uint64_t)d.M * (uint64_t)d.ldx >= (1ull << 31) || (uint64_t)d.w_rows * (uint64_t)d.ldw >= (1ull << 31)) {
      ttv_set_error("gemm (fp32): operand too large for 32-bit element offsets");
      return TTV_ERR_UNSUPPORTED;
    }
    const int nf = ttv_cdiv(d.N, (EPI == EPI_GEGLU) ? 64 : F_TF), nt = ttv_cdiv(d.M, F_TT), nt160 = ttv_cdiv(d.M, 160);
    const long c128 = (long)ttv_cdiv(nf * nt, 512) * 128, c160 = (long)ttv_cdiv(nf * nt160, 512) * 160;
    const bool t160 = ((c160 < c128) && !(d.debug & 256)) || (d.debug & 128);
    // split image on both sides and whole 32-wide k-tiles: both operands by LDS-DMA (TTV_SPLIT3_DMA=0 / ttv_debug_set bit 13: the
    // register-staged kernel, A/B and tests); byte offsets must fit 32 bits
    static const bool s3dma_env = !(getenv("TTV_SPLIT3_DMA") && getenv("TTV_SPLIT3_DMA")[0] == '0');
    const bool s3dma = d.split3 && d.x_image && s3dma_env && !(d.debug & 8192) && d.K % 32 == 0 && d.ldx % 4 == 0 && d.ldw % 4 == 0 &&
                       (uint64_t)d.M * (uint64_t)d.ldx * 4u < (1ull << 32) && (uint64_t)d.w_rows * (uint64_t)d.ldw * 4u < (1ull << 32);
    if (s3dma) {
      if (t160) hipLaunchKernelGGL((k_gemm_split_dma<EPI, 5>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
      else hipLaunchKernelGGL((k_gemm_split_dma<EPI, 4>), dim3(nf * nt), dim3(256), 0, s, d, nf);
    } else if (d.split3) {
      if (t160) hipLaunchKernelGGL((k_gemm_f32<EPI, true, 5>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
      else hipLaunchKernelGGL((k_gemm_f32<EPI, true, 4>), dim3(nf * nt), dim3(256), 0, s, d, nf);
    } else if (s3dma_env && !(d.debug & 8192) && d.K % 32 == 0 && d.ldx % 4 == 0 && d.ldw % 4 == 0 &&
               (uint64_t)d.M * (uint64_t)d.ldx * 4u < (1ull << 32) && (uint64_t)d.w_rows * (uint64_t)d.ldw * 4u < (1ull << 32)) {
      // exact fp32 with both operands by LDS-DMA: the same products in the same order as the register-staged kernel (bit-identical)
      if (t160) hipLaunchKernelGGL((k_gemm_split_dma<EPI, 5, false>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
      else hipLaunchKernelGGL((k_gemm_split_dma<EPI, 4, false>), dim3(nf * nt), dim3(256), 0, s, d, nf);
    } else {
      if (t160) hipLaunchKernelGGL((k_gemm_f32<EPI, false, 5>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
      else hipLaunchKernelGGL((k_gemm_f32<EPI, false, 4>), dim3(nf * nt), dim3(256), 0, s, d, nf);
    }
  }
  TTV_CHECK_LAUNCH("gemm");
  return TTV_OK;
}

// y[M,N] (bf16) = dequant(xq[M,K] e4m3, x_scale[M]) @ dequant(wq[N(or 2I),K] e4m3, w_scale)^T with the EPI_STORE / EPI_QKV_ROPE / EPI_GEGLU epilogues
int ttvk_gemm_fp8(GemmEpilogue epi, const GemmArgs& a, const float* x_scale, const float* w_scale, hipStream_t s, const void* x_mx, const void* w_mx) {
  if (a.M == 0 || a.N == 0) return TTV_OK;
  const bool mx = x_mx || w_mx;
  TTV_CHECK_ARG(epi == EPI_STORE || epi == EPI_QKV_ROPE || epi == EPI_GEGLU || (mx && epi == EPI_RESID_T),
                "gemm_fp8: epilogue must be STORE, QKV_ROPE or GEGLU (block-scaled operands: RESID_T as well)");
  TTV_CHECK_ARG(a.K > 0 && a.K % 128 == 0 && a.N % 8 == 0, "gemm_fp8: K %% 128, N %% 8");
  TTV_CHECK_ARG(a.ldx % 16 == 0 && a.ldw % 16 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.y % 16 == 0),
                "gemm_fp8: 16-byte row alignment");
  if (mx) TTV_CHECK_ARG(x_mx && w_mx && (uintptr_t)x_mx % 4 == 0 && (uintptr_t)w_mx % 4 == 0 && (!w_scale || (uintptr_t)w_scale % 16 == 0),
                        "gemm_fp8: block scales of both operands needed (4-byte aligned)");
  else TTV_CHECK_ARG(x_scale && w_scale && (uintptr_t)w_scale % 16 == 0, "gemm_fp8: scales missing / unaligned");
  if (epi == EPI_RESID_T) TTV_CHECK_ARG(a.resid && a.ldr % 4 == 0, "gemm_fp8: residual missing");
  TTV_CHECK_ARG((uint64_t)a.M * (uint64_t)a.ldx < (1ull << 32), "gemm_fp8: operand too large for 32-bit offsets");
  GemmDev d = {};
  d.x = a.x; d.w = a.w; d.y = a.y; d.bias = a.bias; d.add_scalar = a.add_scalar; d.resid = a.resid; d.rope_cs = a.rope_cs; d.rope_ids = a.rope_ids; d.rope_base = a.rope_base;
  d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = a.ldy; d.ldr = a.ldr; d.M = a.M; d.N = a.N; d.K = a.K; d.alpha = a.alpha;
  d.w_rows = (epi == EPI_GEGLU) ? 2 * a.N : a.N;
  d.rope_q_end = a.rope_q_end; d.rope_k_begin = a.rope_k_begin; d.rope_k_end = a.rope_k_end; d.eps = a.eps; d.debug = g_ttv_debug;
  d.x_scale = x_scale; d.w_scale = w_scale;
  d.x_mx = (const uint8_t*)x_mx; d.w_mx = (const uint8_t*)w_mx; d.ld_mx = 4 * ((a.K / 128 + 3) / 4 * 4);
  if (a.yq) {
    TTV_CHECK_ARG(mx && epi == EPI_GEGLU && a.yq_mx && a.N % 128 == 0 && (uintptr_t)a.yq % 4 == 0, "gemm_fp8: the fp8 GEGLU output needs block-scaled operands, its scale buffer and N %% 128 == 0");
    d.yq = (uint8_t*)a.yq; d.yq_mx = (uint8_t*)a.yq_mx; d.yq_nkp = (a.N / 128 + 3) / 4 * 4; d.ld_yq_mx = 4 * d.yq_nkp;
  }
  if (epi == EPI_QKV_ROPE) TTV_CHECK_ARG(a.rope_cs && a.rope_q_end % 128 == 0 && a.rope_k_begin % 128 == 0 && a.rope_k_end % 128 == 0, "gemm_fp8: rotary ranges must be multiples of 128 columns");
  const int ft = (epi == EPI_GEGLU) ? 64 : TF;
  const int nf = ttv_cdiv(d.N, ft), nt = ttv_cdiv(d.M, TT), nt160 = ttv_cdiv(d.M, 160);
  const long cost128 = (long)ttv_cdiv(nf * nt, 512) * 128, cost160 = (long)ttv_cdiv(nf * nt160, 512) * 160;
  const bool t160 = cost160 < cost128;
  const int kc = epi == EPI_STORE ? TTV_KC_GEMM_STORE : epi == EPI_QKV_ROPE ? TTV_KC_GEMM_QKV : epi == EPI_GEGLU ? TTV_KC_GEMM_GEGLU : TTV_KC_GEMM_RESID;
  TtvProfScope prof(kc, s);
#define F8_LAUNCH(E_, MX_)                                                                                         \
  do {                                                                                                              \
    if (t160) hipLaunchKernelGGL((k_gemm_fp8_dma<E_, 5, MX_>), dim3(nf * nt160), dim3(256), 0, s, d, nf);           \
    else hipLaunchKernelGGL((k_gemm_fp8_dma<E_, 4, MX_>), dim3(nf * nt), dim3(256), 0, s, d, nf);                   \
  } while (0)
  if (mx) {
    if (epi == EPI_STORE) F8_LAUNCH(EPI_STORE, true); else if (epi == EPI_QKV_ROPE) F8_LAUNCH(EPI_QKV_ROPE, true);
    else if (epi == EPI_GEGLU) F8_LAUNCH(EPI_GEGLU, true); else F8_LAUNCH(EPI_RESID_T, true);
  } else {
    if (epi == EPI_STORE) F8_LAUNCH(EPI_STORE, false); else if (epi == EPI_QKV_ROPE) F8_LAUNCH(EPI_QKV_ROPE, false); else F8_LAUNCH(EPI_GEGLU, false);
  }
#undef F8_LAUNCH
  TTV_CHECK_LAUNCH("gemm_fp8");
  return TTV_OK;
}

bool ttvk_gemm_supports_resid_norm(int dtype, int N, int K) { return dtype == TTV_BF16 && N == 256 && K % 8 == 0; }

int ttvk_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (a.M == 0 || a.N == 0) return TTV_OK;
  const int esz = a.dtype == TTV_BF16 ? 2 : 4;
  const int vec = 16 / esz;
  TTV_CHECK_ARG(a.dtype == TTV_BF16 || a.dtype == TTV_F32, "gemm: bad dtype %d", a.dtype);
  TTV_CHECK_ARG(a.K > 0 && a.K % vec == 0, "gemm: K=%d must be a multiple of %d", a.K, vec);
  TTV_CHECK_ARG(a.N % 4 == 0, "gemm: N=%d must be a multiple of 4", a.N);
  TTV_CHECK_ARG(a.ldx % vec == 0 && a.ldw % vec == 0 && a.ldy % vec == 0, "gemm: leading dims must keep 16-byte row alignment");
  TTV_CHECK_ARG(((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.y % 16 == 0), "gemm: pointers must be 16-byte aligned");
  GemmDev d;
  d.x = a.x; d.w = a.w; d.y = a.y; d.bias = a.bias; d.add_scalar = a.add_scalar; d.resid = a.resid; d.rope_cs = a.rope_cs; d.rope_ids = a.rope_ids; d.rope_base = a.rope_base;
  d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = a.ldy; d.ldr = a.ldr;
  d.M = a.M; d.N = a.N; d.K = a.K; d.alpha = a.alpha;
  d.w_rows = (epi == EPI_GEGLU) ? 2 * a.N : a.N;
  d.rope_q_end = a.rope_q_end; d.rope_k_begin = a.rope_k_begin; d.rope_k_end = a.rope_k_end;
  d.eps = a.eps;
  d.debug = g_ttv_debug;
  d.sum_f32 = a.sum_f32; d.ld_sum = a.ld_sum; d.y2 = a.y2; d.ldy2 = a.ldy2; d.norm_gain2 = a.norm_gain2;
  d.split3 = (a.split3 && a.dtype == TTV_F32) ? 1 : 0;
  d.x_image = d.split3 && a.x_image; d.y_image = d.split3 ? a.y_image : 0;
  TTV_CHECK_ARG(!d.y_image || (d.y_image == 1 && (epi == EPI_STORE || epi == EPI_GEGLU)) || (d.y_image == 2 && epi == EPI_QKV_ROPE),
                "gemm: a split-image output is a STORE / GEGLU option (y_image 1) or the to_qkv layout (y_image 2)");
  d.x_mx = d.w_mx = nullptr; d.ld_mx = 0;
  d.norm_gain = a.norm_gain;
  d.clip_desc = a.clip_desc; d.patch_rows = a.patch_rows; d.row_seq = a.row_seq; d.clip0 = 0; d.pt_shift = d.ph_shift = 0;
  d.x_rows = a.x_rows;
  d.x_scale = nullptr; d.w_scale = nullptr;
  d.row_scale = a.row_scale;
  d.stamps = g_ttv_stamps;
  TTV_CHECK_ARG(!a.row_scale || ((a.dtype == TTV_BF16 || a.split3) && (epi == EPI_STORE || epi == EPI_QKV_ROPE || epi == EPI_GEGLU)),
                "gemm: row_scale is a STORE / QKV_ROPE / GEGLU option of the bf16 and the split-bf16 kernels");
  TTV_CHECK_ARG(!a.x_rows || (a.dtype == TTV_BF16 && a.K == 256 && a.N % 8 == 0 && epi != EPI_RESID_NORM), "gemm: x_rows needs the bf16 K=256 kernel");
  if (epi == EPI_STORE_PATCH || a.gather) {
    auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    TTV_CHECK_ARG(a.dtype == TTV_BF16 && (a.gather || a.K == 256) && a.patch_w == 8 && lg2(a.patch_t) >= 0 && lg2(a.patch_h) >= 0,
                  "gemm: patch gather/scatter needs bf16, patch_w == 8 and power-of-two patch_t / patch_h (scatter: K == 256)");
    TTV_CHECK_ARG(!a.gather || (epi == EPI_STORE && a.K % 64 == 0 && a.K != 256), "gemm: patch gather is an EPI_STORE option of the general-K kernel");
    TTV_CHECK_ARG(a.clips && a.n_clips > 0 && a.n_clips <= TTV_MAX_CLIPS_PER_LAUNCH && a.clip_desc && a.patch_rows && a.row_seq,
                  "gemm: patch scatter needs clips (<= %d), clip_desc, patch_rows and row_seq", TTV_MAX_CLIPS_PER_LAUNCH);
    for (int i = 0; i < a.n_clips; ++i) d.clips.p[i] = a.clips[i];
    d.pt_shift = lg2(a.patch_t); d.ph_shift = lg2(a.patch_h);
  }
  const bool pn = a.prenorm != 0;
  const int kc = (epi == EPI_STORE || epi == EPI_STORE_PATCH) ? TTV_KC_GEMM_STORE : epi == EPI_QKV_ROPE ? TTV_KC_GEMM_QKV : epi == EPI_GEGLU ? TTV_KC_GEMM_GEGLU : TTV_KC_GEMM_RESID;
  TtvProfScope prof(kc, s);
  switch (epi) {
    case EPI_STORE:
      if (a.gather) {
        const int nf = ttv_cdiv(d.N, TF), nt = ttv_cdiv(d.M, TT), nt160 = ttv_cdiv(d.M, 160);
        const int force_tt = (d.debug & 128) ? 160 : (d.debug & 256) ? 128 : 0;
        const long cost128 = (long)ttv_cdiv(nf * nt, 512) * 128, cost160 = (long)ttv_cdiv(nf * nt160, 512) * 160;
        if ((cost160 < cost128 && force_tt != 128) || force_tt == 160)
          hipLaunchKernelGGL((k_gemm_bf16<EPI_STORE, true, 5>), dim3(nf * nt160), dim3(256), 0, s, d, nf);
        else
          hipLaunchKernelGGL((k_gemm_bf16<EPI_STORE, true>), dim3(nf * nt), dim3(256), 0, s, d, nf);
        TTV_CHECK_LAUNCH("gemm_gather");
        return TTV_OK;
      }
      return launch<EPI_STORE>(d, a.dtype, pn, s);
    case EPI_STORE_PATCH: return launch<EPI_STORE_PATCH>(d, a.dtype, pn, s);
    case EPI_QKV_ROPE:
      // rotary ranges must be multiples of the largest feature tile (128) so that "rotate or not" is uniform per block
      TTV_CHECK_ARG(a.rope_cs && a.rope_q_end % 128 == 0 && a.rope_k_begin % 128 == 0 && a.rope_k_end % 128 == 0, "gemm: rotary ranges must be multiples of 128 columns");
      return launch<EPI_QKV_ROPE>(d, a.dtype, pn, s);
    case EPI_GEGLU: return launch<EPI_GEGLU>(d, a.dtype, pn, s);
    case EPI_RESID_T:
      TTV_CHECK_ARG(a.resid && a.ldr % 4 == 0, "gemm: residual missing");
      return launch<EPI_RESID_T>(d, a.dtype, pn, s);
    case EPI_RESID_F32:
      TTV_CHECK_ARG(a.resid && a.ldr % 4 == 0, "gemm: residual missing");
      return launch<EPI_RESID_F32>(d, a.dtype, pn, s);
    case EPI_RESID_NORM: {
      TTV_CHECK_ARG(ttvk_gemm_supports_resid_norm(a.dtype, a.N, a.K), "gemm: fused residual+norm needs bf16 and N == 256");
      TTV_CHECK_ARG(a.resid && a.norm_gain && a.ldr % 4 == 0 && a.ldw >= 256, "gemm: residual / gain missing");
      TTV_CHECK_ARG(a.y != a.x, "gemm: fused residual+norm output must not alias the GEMM input");
      TTV_CHECK_ARG(!a.y2 || (a.norm_gain2 && a.ldy2 % 8 == 0 && (uintptr_t)a.y2 % 16 == 0 && a.y2 != a.x), "gemm: second norm output needs its gain, 16-byte rows");
      TTV_CHECK_ARG(!a.sum_f32 || (a.ld_sum % 4 == 0 && (uintptr_t)a.sum_f32 % 16 == 0), "gemm: fp32 sum output needs 16-byte rows");
      if (a.K == 256) {
        const int n_tiles = ttv_cdiv(a.M, ROW_TT);
        hipLaunchKernelGGL(k_gemm_k256_rownorm, dim3(n_tiles < 512 ? n_tiles : 512), dim3(256), 0, s, d, n_tiles);
        TTV_CHECK_LAUNCH("gemm_k256_rownorm");
      } else {
        hipLaunchKernelGGL(k_gemm_rowtile_norm, dim3(ttv_cdiv(a.M, 64)), dim3(256), 0, s, d);
        TTV_CHECK_LAUNCH("gemm_rowtile_norm");
      }
      return TTV_OK;
    }
  }
  ttv_set_error("gemm: unknown epilogue");
  return TTV_ERR_INVALID;
}

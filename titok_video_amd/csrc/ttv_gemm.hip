// Linear layers of the TiTok-Video towers (proj_in/out, to_qkv, out_proj, w12, w3) as y = x @ w^T.
//
// Orientation: the MFMA "row" (A) operand is the WEIGHT tile (output features), the "column" (B) operand the
// TOKEN tile, i.e. the kernel computes y^T = w @ x^T.  With mfma_f32_16x16x32_bf16's C layout
// (col = lane&15, row = 4*(lane>>4)+reg) a lane then owns 4 CONSECUTIVE output features of one token, so
// rotary pairs, GEGLU's (x, gate) pair, bias and the residual are lane-local and the store is one 8-byte vector.
// Both operands are K-contiguous in memory ([N,K] weights, [M,K] activations), which is what MFMA wants.
//
// bf16 path : 128 features x 128 tokens x 64 K per step, 4 waves (2x2), each 64x64 = 4x4 MFMA tiles,
//             register-staged double-buffered LDS, XOR-swizzled 16-byte chunks (conflict-free ds_read_b128),
//             XCD-aware tile order (feature tile fastest so the token tile is re-read from the same L2).
// fp32 path : parity instrument only (the reference never runs fp32 on GPU): 64x64x16 tiles, VALU FMAs.
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
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// Lane-local epilogue on 4 consecutive features [f0, f0+4) of token t.  acc2 is the gate half for EPI_GEGLU.
template <int EPI, typename T>
__device__ __forceinline__ void epilogue(const GemmDev& p, int t, int f0, f32x4 acc, f32x4 acc2) {
  if (t >= p.M || f0 >= p.N) return;
  if (EPI == EPI_STORE) {
    if (p.bias) {
      const f32x4 b = Vec4<T>::load((const T*)p.bias + f0);
      acc += b;
    }
    if (p.add_scalar) {
      const float sc = round_to<T>(p.add_scalar[0]);
      if (p.bias) {  // Linear output is rounded to dtype before the scalar is added (blocks.py:97)
        acc = (f32x4){round_to<T>(acc[0]), round_to<T>(acc[1]), round_to<T>(acc[2]), round_to<T>(acc[3])};
      }
      acc += sc;
    }
    Vec4<T>::store((T*)p.y + (size_t)t * p.ldy + f0, acc);
  } else if (EPI == EPI_QKV_ROPE) {
    const bool rot = (f0 < p.rope_q_end) || (f0 >= p.rope_k_begin && f0 < p.rope_k_end);
    if (rot) {
      const int j = (f0 & 63) >> 1;
      const float* cs = p.rope_cs + (size_t)t * 64 + j;
      const float c0 = cs[0], c1 = cs[1], s0 = cs[32], s1 = cs[33];
      acc = (f32x4){acc[0] * c0 - acc[1] * s0, acc[0] * s0 + acc[1] * c0, acc[2] * c1 - acc[3] * s1, acc[2] * s1 + acc[3] * c1};
    }
    Vec4<T>::store((T*)p.y + (size_t)t * p.ldy + f0, acc);
  } else if (EPI == EPI_GEGLU) {
    f32x4 o = {gelu_erf(acc2[0]) * acc[0], gelu_erf(acc2[1]) * acc[1], gelu_erf(acc2[2]) * acc[2], gelu_erf(acc2[3]) * acc[3]};
    Vec4<T>::store((T*)p.y + (size_t)t * p.ldy + f0, o);
  } else if (EPI == EPI_RESID_T) {
    const f32x4 r = Vec4<T>::load((const T*)p.resid + (size_t)t * p.ldr + f0);
    acc += p.alpha * r;
    Vec4<T>::store((T*)p.y + (size_t)t * p.ldy + f0, acc);
  } else {  // EPI_RESID_F32
    const f32x4 r = Vec4<T>::load((const T*)p.resid + (size_t)t * p.ldr + f0);
    acc += p.alpha * r;
    *reinterpret_cast<f32x4*>((float*)p.y + (size_t)t * p.ldy + f0) = acc;
  }
}

// ================================================================================================
// bf16 MFMA kernel
// ================================================================================================
#define TF 128
#define TT 128
#define BK 64

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  constexpr int FT = DUAL ? 64 : TF;  // output features per block
  __shared__ uint4 lds[2][2][TF * 8];  // [buffer][operand: 0 = w, 1 = x][row*8 + swizzled chunk], 64 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wf = wave & 1, wt = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int fbase = (tile % n_ftiles) * FT;
  const int tbase = (tile / n_ftiles) * TT;

  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;

  // per-thread staging: 4 chunks (16 B) of each operand per k-tile; chunk = tid + 256*i -> row = chunk>>3, kc = chunk&7
  // (row advances by 32 per i, kc is the same for all i)
  const int srow = tid >> 3, skc = tid & 7;
  const bf16_t* wp0; const bf16_t* wp1; const bf16_t* wp2; const bf16_t* wp3;
  const bf16_t* xp0; const bf16_t* xp1; const bf16_t* xp2; const bf16_t* xp3;
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
  }
  // swizzled LDS slot of (row, kc): row*8 + (kc ^ ((row>>1)&7)); rows srow+32*i share (row>>1)&7 up to +16*i
  const int li0 = srow * 8 + (skc ^ ((srow >> 1) & 7));
  const int li1 = (srow + 32) * 8 + (skc ^ (((srow + 32) >> 1) & 7));
  const int li2 = (srow + 64) * 8 + (skc ^ (((srow + 64) >> 1) & 7));
  const int li3 = (srow + 96) * 8 + (skc ^ (((srow + 96) >> 1) & 7));

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 sw0, sw1, sw2, sw3, sx0, sx1, sx2, sx3;
  const uint4 zero4 = {0u, 0u, 0u, 0u};
#define GLOAD(k0)                                                                   \
  do {                                                                              \
    const bool ok__ = ((k0) + skc * 8) < p.K;                                       \
    sw0 = ok__ ? *reinterpret_cast<const uint4*>(wp0 + (k0)) : zero4;               \
    sw1 = ok__ ? *reinterpret_cast<const uint4*>(wp1 + (k0)) : zero4;               \
    sw2 = ok__ ? *reinterpret_cast<const uint4*>(wp2 + (k0)) : zero4;               \
    sw3 = ok__ ? *reinterpret_cast<const uint4*>(wp3 + (k0)) : zero4;               \
    sx0 = ok__ ? *reinterpret_cast<const uint4*>(xp0 + (k0)) : zero4;               \
    sx1 = ok__ ? *reinterpret_cast<const uint4*>(xp1 + (k0)) : zero4;               \
    sx2 = ok__ ? *reinterpret_cast<const uint4*>(xp2 + (k0)) : zero4;               \
    sx3 = ok__ ? *reinterpret_cast<const uint4*>(xp3 + (k0)) : zero4;               \
  } while (0)
#define LSTORE(buf)                                                                 \
  do {                                                                              \
    lds[buf][0][li0] = sw0; lds[buf][0][li1] = sw1; lds[buf][0][li2] = sw2; lds[buf][0][li3] = sw3; \
    lds[buf][1][li0] = sx0; lds[buf][1][li1] = sx1; lds[buf][1][li2] = sx2; lds[buf][1][li3] = sx3; \
  } while (0)

  // fragment rows: A (weights) rows of this wave's 4 m-tiles, B (tokens) rows of its 4 n-tiles
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
      bf16x8 a[4], b[4];
      const int kc = ks * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = (DUAL ? (i < 2 ? wf * 32 + i * 16 : 64 + wf * 32 + (i - 2) * 16) : wf * 64 + i * 16) + l15;
        const int brow = wt * 64 + i * 16 + l15;
        a[i] = __builtin_bit_cast(bf16x8, lds[buf][0][arow * 8 + (kc ^ ((arow >> 1) & 7))]);
        b[i] = __builtin_bit_cast(bf16x8, lds[buf][1][brow * 8 + (kc ^ ((brow >> 1) & 7))]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef GLOAD
#undef LSTORE

  // epilogue: lane owns features f0..f0+3 (4*(lane>>4)+reg) of token (lane&15) in every 16x16 tile
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tbase + wt * 64 + j * 16 + (lane & 15);
    if (DUAL) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f0 = fbase + wf * 32 + i * 16 + (lane >> 4) * 4;
        epilogue<EPI, bf16_t>(p, t, f0, acc[i][j], acc[i + 2][j]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f0 = fbase + wf * 64 + i * 16 + (lane >> 4) * 4;
        epilogue<EPI, bf16_t>(p, t, f0, acc[i][j], acc[i][j]);
      }
    }
  }
}

// ================================================================================================
// bf16 MFMA kernel for K == 256 (every to_qkv / w12 / out_proj / decoder proj_out of a width-256 tower)
//
// With K this short a classic tiled GEMM spends its time in per-tile load latency, so this kernel keeps the TOKEN
// operand in registers for the whole K (a wave owns 64 tokens: 4 n-tiles x 8 k-steps x 16 B = 128 VGPRs) and streams
// 64-row WEIGHT panels (32 KiB) through double-buffered LDS.  Blocks are persistent over a contiguous range of
// (token tile, panel) items, so the token registers are loaded once per 128-token tile and the steady state is
// {prefetch next panel -> 64 MFMA per wave from LDS -> epilogue -> stage -> one barrier}.
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

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wf = wave & 1, wt = wave >> 1;
  const int l15 = lane & 15, kq = lane >> 4;
  const int it0 = (int)((long)total_items * blockIdx.x / gridDim.x);
  const int it1 = (int)((long)total_items * (blockIdx.x + 1) / gridDim.x);
  if (it0 >= it1) return;

  const bf16_t* W = (const bf16_t*)p.w;
  const bf16_t* X = (const bf16_t*)p.x;

  // staging: 8 chunks per thread per panel; chunk = tid + 256*i -> row = (tid>>5) + 8*i, ch = tid & 31
  const int srow = tid >> 5, sch = tid & 31;
  uint4 st0, st1, st2, st3, st4, st5, st6, st7;
#define WROW(panel_, i_)                                                                                     \
  ({                                                                                                         \
    const int row__ = srow + 8 * (i_);                                                                       \
    int wr__ = DUAL ? (row__ < 32 ? (panel_) * 32 + row__ : p.N + (panel_) * 32 + (row__ - 32)) : (panel_) * 64 + row__; \
    wr__ = wr__ < p.w_rows ? wr__ : p.w_rows - 1;                                                            \
    *reinterpret_cast<const uint4*>(W + (size_t)wr__ * p.ldw + sch * 8);                                     \
  })
#define GLOADP(panel_)                                                                                       \
  do {                                                                                                       \
    st0 = WROW(panel_, 0); st1 = WROW(panel_, 1); st2 = WROW(panel_, 2); st3 = WROW(panel_, 3);              \
    st4 = WROW(panel_, 4); st5 = WROW(panel_, 5); st6 = WROW(panel_, 6); st7 = WROW(panel_, 7);              \
  } while (0)
#define LIDX(i_) ((srow + 8 * (i_)) * 32 + ((sch & 16) | ((sch & 15) ^ ((srow + 8 * (i_)) & 15))))
#define LSTOREP(buf_)                                                                                        \
  do {                                                                                                       \
    wl[buf_][LIDX(0)] = st0; wl[buf_][LIDX(1)] = st1; wl[buf_][LIDX(2)] = st2; wl[buf_][LIDX(3)] = st3;      \
    wl[buf_][LIDX(4)] = st4; wl[buf_][LIDX(5)] = st5; wl[buf_][LIDX(6)] = st6; wl[buf_][LIDX(7)] = st7;      \
  } while (0)

  // A-fragment rows of this wave inside a panel
  const int arow0 = (DUAL ? wf * 16 : wf * 32) + l15;
  const int arow1 = (DUAL ? 32 + wf * 16 : wf * 32 + 16) + l15;

  bf16x8 bfr[4][8];
  float rstd[4] = {1.f, 1.f, 1.f, 1.f};
  int cur_tile = -1;

  GLOADP(it0 % n_panels);
  LSTOREP(0);
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) & 1;
    const int tile = it / n_panels, panel = it - tile * n_panels;
    if (tile != cur_tile) {
      cur_tile = tile;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int t = tile * K256_TT + wt * 64 + j * 16 + l15;
        t = t < p.M ? t : p.M - 1;
        const bf16_t* xr = X + (size_t)t * p.ldx + kq * 8;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) bfr[j][s8] = *reinterpret_cast<const bf16x8*>(xr + s8 * 32);
      }
      if (PRENORM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float ss = 0.f;
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = (float)bfr[j][s8][e];
              ss = fmaf(v, v, ss);
            }
          ss += __shfl_xor(ss, 16, 64);
          ss += __shfl_xor(ss, 32, 64);
          rstd[j] = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
        }
      }
    }
    if (it + 1 < it1) GLOADP((it + 1) % n_panels);

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int ch = s8 * 4 + kq;
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, wl[buf][arow0 * 32 + ((ch & 16) | ((ch & 15) ^ (arow0 & 15)))]);
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, wl[buf][arow1 * 32 + ((ch & 16) | ((ch & 15) ^ (arow1 & 15)))]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bfr[j][s8], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bfr[j][s8], acc[1][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = tile * K256_TT + wt * 64 + j * 16 + l15;
      if (PRENORM) {
        acc[0][j] *= rstd[j];
        acc[1][j] *= rstd[j];
      }
      if (DUAL) {
        epilogue<EPI, bf16_t>(p, t, panel * FO + wf * 16 + kq * 4, acc[0][j], acc[1][j]);
      } else {
        epilogue<EPI, bf16_t>(p, t, panel * FO + wf * 32 + kq * 4, acc[0][j], acc[0][j]);
        epilogue<EPI, bf16_t>(p, t, panel * FO + wf * 32 + 16 + kq * 4, acc[1][j], acc[1][j]);
      }
    }
    if (it + 1 < it1) LSTOREP(buf ^ 1);
    __syncthreads();
  }
#undef WROW
#undef GLOADP
#undef LIDX
#undef LSTOREP
}

// ================================================================================================
// fp32 kernel (parity instrument)
// ================================================================================================
#define F_TF 64
#define F_TT 64
#define F_BK 16

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmDev p, int n_ftiles) {
  constexpr bool DUAL = (EPI == EPI_GEGLU);
  __shared__ float ws[F_BK][F_TF + 4];
  __shared__ float ws2[DUAL ? F_BK : 1][F_TF + 4];
  __shared__ float xs[F_BK][F_TT + 4];
  const int tid = threadIdx.x;
  const int fbase = (blockIdx.x % n_ftiles) * F_TF;
  const int tbase = (blockIdx.x / n_ftiles) * F_TT;
  const float* W = (const float*)p.w;
  const float* X = (const float*)p.x;
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  int wr = fbase + lrow; wr = wr < p.N ? wr : p.N - 1;
  int xr = tbase + lrow; xr = xr < p.M ? xr : p.M - 1;
  const int tf = tid & 15, tt = tid >> 4;
  float acc[4][4] = {}, acc2[4][4] = {};
  for (int k0 = 0; k0 < p.K; k0 += F_BK) {
    const bool ok = (k0 + lk) < p.K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 vw = ok ? *reinterpret_cast<const f32x4*>(W + (size_t)wr * p.ldw + k0 + lk) : zero;
    const f32x4 vx = ok ? *reinterpret_cast<const f32x4*>(X + (size_t)xr * p.ldx + k0 + lk) : zero;
    f32x4 vw2 = zero;
    if (DUAL) vw2 = ok ? *reinterpret_cast<const f32x4*>(W + (size_t)(p.N + wr) * p.ldw + k0 + lk) : zero;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ws[lk + e][lrow] = vw[e];
      xs[lk + e][lrow] = vx[e];
      if (DUAL) ws2[lk + e][lrow] = vw2[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < F_BK; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&ws[k][tf * 4]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&xs[k][tt * 4]);
      f32x4 a2 = a;
      if (DUAL) a2 = *reinterpret_cast<const f32x4*>(&ws2[k][tf * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
          if (DUAL) acc2[i][j] = fmaf(a2[i], b[j], acc2[i][j]);
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tbase + tt * 4 + j;
    const f32x4 v = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
    const f32x4 v2 = {acc2[0][j], acc2[1][j], acc2[2][j], acc2[3][j]};
    epilogue<EPI, float>(p, t, fbase + tf * 4, v, v2);
  }
}

template <int EPI>
static int launch(const GemmDev& d, int dtype, bool prenorm, hipStream_t s) {
  if (dtype == TTV_BF16 && d.K == 256) {
    const int fo = (EPI == EPI_GEGLU) ? 32 : 64;
    const int n_panels = ttv_cdiv(d.N, fo), n_tiles = ttv_cdiv(d.M, K256_TT);
    const int total = n_panels * n_tiles;
    const int grid = total < 512 ? total : 512;   // 2 co-resident blocks per CU
    if (prenorm) hipLaunchKernelGGL((k_gemm_k256<EPI, true>), dim3(grid), dim3(256), 0, s, d, n_panels, total);
    else hipLaunchKernelGGL((k_gemm_k256<EPI, false>), dim3(grid), dim3(256), 0, s, d, n_panels, total);
    TTV_CHECK_LAUNCH("gemm_k256");
    return TTV_OK;
  }
  if (prenorm) {
    ttv_set_error("gemm: folded pre-norm needs the bf16 K=256 kernel");
    return TTV_ERR_UNSUPPORTED;
  }
  if (dtype == TTV_BF16) {
    const int ft = (EPI == EPI_GEGLU) ? 64 : TF;
    const int nf = ttv_cdiv(d.N, ft), nt = ttv_cdiv(d.M, TT);
    hipLaunchKernelGGL((k_gemm_bf16<EPI>), dim3(nf * nt), dim3(256), 0, s, d, nf);
  } else {
    const int nf = ttv_cdiv(d.N, F_TF), nt = ttv_cdiv(d.M, F_TT);
    hipLaunchKernelGGL((k_gemm_f32<EPI>), dim3(nf * nt), dim3(256), 0, s, d, nf);
  }
  TTV_CHECK_LAUNCH("gemm");
  return TTV_OK;
}

int ttvk_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (a.M == 0 || a.N == 0) return TTV_OK;
  const int esz = a.dtype == TTV_BF16 ? 2 : 4;
  const int vec = 16 / esz;
  TTV_CHECK_ARG(a.dtype == TTV_BF16 || a.dtype == TTV_F32, "gemm: bad dtype %d", a.dtype);
  TTV_CHECK_ARG(a.K > 0 && a.K % vec == 0, "gemm: K=%d must be a multiple of %d", a.K, vec);
  TTV_CHECK_ARG(a.N % 4 == 0, "gemm: N=%d must be a multiple of 4", a.N);
  TTV_CHECK_ARG(a.ldx % vec == 0 && a.ldw % vec == 0 && a.ldy % 4 == 0, "gemm: leading dims must keep 16-byte row alignment");
  TTV_CHECK_ARG(((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.y % 8 == 0), "gemm: pointers must be 16-byte aligned");
  GemmDev d;
  d.x = a.x; d.w = a.w; d.y = a.y; d.bias = a.bias; d.add_scalar = a.add_scalar; d.resid = a.resid; d.rope_cs = a.rope_cs;
  d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = a.ldy; d.ldr = a.ldr;
  d.M = a.M; d.N = a.N; d.K = a.K; d.alpha = a.alpha;
  d.w_rows = (epi == EPI_GEGLU) ? 2 * a.N : a.N;
  d.rope_q_end = a.rope_q_end; d.rope_k_begin = a.rope_k_begin; d.rope_k_end = a.rope_k_end;
  d.eps = a.eps;
  const bool pn = a.prenorm != 0;
  const int kc = epi == EPI_STORE ? TTV_KC_GEMM_STORE : epi == EPI_QKV_ROPE ? TTV_KC_GEMM_QKV : epi == EPI_GEGLU ? TTV_KC_GEMM_GEGLU : TTV_KC_GEMM_RESID;
  TtvProfScope prof(kc, s);
  switch (epi) {
    case EPI_STORE: return launch<EPI_STORE>(d, a.dtype, pn, s);
    case EPI_QKV_ROPE:
      TTV_CHECK_ARG(a.rope_cs && a.rope_q_end % 64 == 0 && a.rope_k_begin % 64 == 0 && a.rope_k_end % 64 == 0, "gemm: rotary ranges must be head (64) aligned");
      return launch<EPI_QKV_ROPE>(d, a.dtype, pn, s);
    case EPI_GEGLU: return launch<EPI_GEGLU>(d, a.dtype, pn, s);
    case EPI_RESID_T:
      TTV_CHECK_ARG(a.resid && a.ldr % 4 == 0, "gemm: residual missing");
      return launch<EPI_RESID_T>(d, a.dtype, pn, s);
    case EPI_RESID_F32:
      TTV_CHECK_ARG(a.resid && a.ldr % 4 == 0, "gemm: residual missing");
      return launch<EPI_RESID_F32>(d, a.dtype, pn, s);
  }
  ttv_set_error("gemm: unknown epilogue");
  return TTV_ERR_INVALID;
}

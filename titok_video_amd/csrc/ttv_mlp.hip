// Fused GEGLU feed-forward sub-layer of a width-256 tower, one kernel (reference model/base/transformer.py:47-56 + the
// residual / KEEL step at :130 and :144-145):
//
//     h     = gelu_erf(x_n W12g^T) * (x_n W12x^T)          x_n = RMSNorm(x) * norm.weight   (folded, see below)
//     y     = alpha * x + h W3^T
//     x_new = KEEL ? RMSNorm(y) * post_gain : y
//
// Why one kernel: unfused, this sub-layer moves x (19 MB) -> h (52 MB write + 52 MB read) -> y fp32 (38 MB write + read)
// -> x at the benchmark shape; fused it reads x once and writes x_new once (38 MB) and h never leaves the CU.
//
// Structure (bf16, K = d = 256, I % 32 == 0):
//   * a wave owns NT*16 tokens: their x rows stay in registers as MFMA B fragments for the whole kernel
//     (NT x 8 k-steps x 16 B), the pre-norm is folded (gain pre-multiplied into W12's columns on the host, rstd from the
//     register-resident row), and the wave accumulates ALL 256 output features of its tokens (16 m-tiles x NT), so the
//     KEEL RMSNorm is wave-local (in-lane squares + two xor shuffles).
//   * the hidden dimension is walked in panels of 32 (x, gate) pairs: a 64-row W12 panel (32 KiB) and the matching
//     256 x 32 W3 panel (16 KiB) stream L2 -> registers -> double-buffered LDS, shared by the block's 4 waves.
//   * phase 1: 4 m-tiles (x0, x1, g0, g1) x NT x 8 k-steps MFMAs -> GEGLU in registers.  With mfma_16x16x32's C layout a lane
//     then holds, for its token, pairs {4kq..4kq+3} and {16+4kq..16+4kq+3}: exactly 8 values = one B fragment of the
//     second product, provided the k order of that product is  k(kq, j) = j<4 ? 4kq+j : 16+4kq+(j-4).  W3's columns are
//     stored in that order on the host (weights.py), so phase 2 needs NO cross-lane movement and no LDS round trip.
//   * phase 2: 16 m-tiles x NT MFMAs (one 32-deep k-step) accumulate y^T.
//   * epilogue: residual reload (L2-hot), alpha*x + acc, row statistics, gain, lane-pair exchange, 16-byte stores.
#include "ttv_common.h"
#include "ttv_kernels.h"

struct MlpDev {
  const bf16_t* x; int ldx;
  const bf16_t* w12;   // [2I, 256], pre-norm gain folded into the columns
  const bf16_t* w3p;   // [256, I], columns permuted per 32-pair panel (see above)
  int I;
  bf16_t* y; int ldy;
  const float* post_gain;
  float alpha, eps;
  int M, n_tiles, debug;
};

template <int NT, bool KEEL>
__global__ __launch_bounds__(256, 1) void k_mlp256(MlpDev p) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* l12 = smem;                  // [2][64 rows * 32 chunks]   W12 panel, XOR-swizzled chunks
  uint4* l3 = smem + 2 * 64 * 32;     // [2][4 chunks][256 rows]    W3 panel, chunk-major (conflict-free b128 reads)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int np = p.I / 32;

  // staging assignments
  const int srow = tid >> 5, sch = tid & 31;   // W12: rows srow + 8i (i < 8), 16-byte chunk sch
  uint4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3;
#define W12ROW(pn_, i_)                                                                                            \
  ({                                                                                                               \
    const int row__ = srow + 8 * (i_);                                                                             \
    const int wr__ = row__ < 32 ? (pn_) * 32 + row__ : p.I + (pn_) * 32 + (row__ - 32);                            \
    *reinterpret_cast<const uint4*>(p.w12 + (size_t)wr__ * 256 + sch * 8);                                         \
  })
#define GLOAD_PANELS(pn_)                                                                                          \
  do {                                                                                                             \
    a0 = W12ROW(pn_, 0); a1 = W12ROW(pn_, 1); a2 = W12ROW(pn_, 2); a3 = W12ROW(pn_, 3);                            \
    a4 = W12ROW(pn_, 4); a5 = W12ROW(pn_, 5); a6 = W12ROW(pn_, 6); a7 = W12ROW(pn_, 7);                            \
    const bf16_t* w3r__ = p.w3p + (size_t)tid * p.I + (pn_) * 32;                                                  \
    b0 = *reinterpret_cast<const uint4*>(w3r__); b1 = *reinterpret_cast<const uint4*>(w3r__ + 8);                  \
    b2 = *reinterpret_cast<const uint4*>(w3r__ + 16); b3 = *reinterpret_cast<const uint4*>(w3r__ + 24);            \
  } while (0)
#define L12IDX(i_) ((srow + 8 * (i_)) * 32 + ((sch & 16) | ((sch & 15) ^ ((srow + 8 * (i_)) & 15))))
#define LSTORE_PANELS(buf_)                                                                                        \
  do {                                                                                                             \
    uint4* d12__ = l12 + (buf_) * (64 * 32);                                                                       \
    d12__[L12IDX(0)] = a0; d12__[L12IDX(1)] = a1; d12__[L12IDX(2)] = a2; d12__[L12IDX(3)] = a3;                    \
    d12__[L12IDX(4)] = a4; d12__[L12IDX(5)] = a5; d12__[L12IDX(6)] = a6; d12__[L12IDX(7)] = a7;                    \
    uint4* d3__ = l3 + (buf_) * (4 * 256);                                                                         \
    d3__[0 * 256 + tid] = b0; d3__[1 * 256 + tid] = b1; d3__[2 * 256 + tid] = b2; d3__[3 * 256 + tid] = b3;        \
  } while (0)

  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    // ---- token rows of this wave: B fragments + folded pre-norm rstd ----
    int tok[NT];
    bf16x8 bfr[NT][8];
    float rstd[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      tok[j] = tile * (64 * NT) + wave * (16 * NT) + j * 16 + l15;
      const int tc = tok[j] < p.M ? tok[j] : p.M - 1;
      const bf16_t* xr = p.x + (size_t)tc * p.ldx + kq * 8;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) bfr[j][s8] = *reinterpret_cast<const bf16x8*>(xr + s8 * 32);
    }
    GLOAD_PANELS(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
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
    __syncthreads();   // previous tile's last panel fully consumed before buffer 0 is overwritten
    LSTORE_PANELS(0);
    __syncthreads();

    f32x4 out[16][NT];
#pragma unroll
    for (int m = 0; m < 16; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j) out[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int pn = 0; pn < np; ++pn) {
      const int buf = pn & 1;
      if (pn + 1 < np && !(p.debug & 2)) GLOAD_PANELS(pn + 1);
      const uint4* w12l = l12 + buf * (64 * 32);
      const uint4* w3l = l3 + buf * (4 * 256);

      // ---- phase 1: (x0, x1, g0, g1) m-tiles of this panel ----
      f32x4 acc1[4][NT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int ch = s8 * 4 + kq;
        bf16x8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int arow = i * 16 + l15;
          a[i] = __builtin_bit_cast(bf16x8, w12l[arow * 32 + ((ch & 16) | ((ch & 15) ^ (arow & 15)))]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bfr[j][s8], acc1[i][j], 0, 0, 0);
      }
      // ---- GEGLU in registers -> B fragments of phase 2 ----
      bf16x8 hf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float hv[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 gg = (f32x2){acc1[2 + i][j][e], acc1[2 + i][j][e + 1]} * rstd[j];
            const f32x2 xx = (f32x2){acc1[i][j][e], acc1[i][j][e + 1]} * rstd[j];
            const f32x2 h2 = geglu_pair_fast(gg, xx);
            hv[4 * i + e] = h2.x;
            hv[4 * i + e + 1] = h2.y;
          }
        hf[j] = (bf16x8){(bf16_t)hv[0], (bf16_t)hv[1], (bf16_t)hv[2], (bf16_t)hv[3],
                         (bf16_t)hv[4], (bf16_t)hv[5], (bf16_t)hv[6], (bf16_t)hv[7]};
      }
      // ---- phase 2: y^T += W3panel h^T ----
      // A fragments are fetched 8 at a time ahead of their MFMAs (one LDS latency per 8 m-tiles, not one per m-tile)
#pragma unroll
      for (int mg = 0; mg < 2; ++mg) {
        bf16x8 a3f[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a3f[m] = __builtin_bit_cast(bf16x8, w3l[kq * 256 + (mg * 8 + m) * 16 + l15]);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j) out[mg * 8 + m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3f[m], hf[j], out[mg * 8 + m][j], 0, 0, 0);
      }
      if (pn + 1 < np && !(p.debug & 2)) LSTORE_PANELS(buf ^ 1);
      __syncthreads();
    }

    // ---- epilogue: y = alpha*x + acc ; x_new = KEEL ? RMSNorm(y)*gain : y ----
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bool tv = tok[j] < p.M;
      const int tc = tv ? tok[j] : p.M - 1;
      const bf16_t* rrow = p.x + (size_t)tc * p.ldx + kq * 4;
      f32x4 r[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) r[m] = Vec4<bf16_t>::load(rrow + m * 16);
      float ss = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        out[m][j] += p.alpha * r[m];
#pragma unroll
        for (int e = 0; e < 4; ++e) ss = fmaf(out[m][j][e], out[m][j][e], ss);
      }
      float scale = 1.0f;
      if (KEEL) {
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        scale = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
      }
      const bool odd = kq & 1;
      bf16_t* yrow = p.y + (size_t)tc * p.ldy;
#pragma unroll
      for (int ip = 0; ip < 8; ++ip) {
        const int i0 = 2 * ip, i1 = 2 * ip + 1;
        f32x4 y0 = out[i0][j], y1 = out[i1][j];
        if (KEEL) {
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.post_gain + i0 * 16 + kq * 4);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.post_gain + i1 * 16 + kq * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { y0[e] = y0[e] * scale * g0[e]; y1[e] = y1[e] * scale * g1[e]; }
        }
        const bf16x4 q0 = {(bf16_t)y0[0], (bf16_t)y0[1], (bf16_t)y0[2], (bf16_t)y0[3]};
        const bf16x4 q1 = {(bf16_t)y1[0], (bf16_t)y1[1], (bf16_t)y1[2], (bf16_t)y1[3]};
        const uint2 p0 = __builtin_bit_cast(uint2, q0), p1 = __builtin_bit_cast(uint2, q1);
        const uint2 send = odd ? p0 : p1;
        uint2 recv;
        recv.x = __shfl_xor(send.x, 16, 64);
        recv.y = __shfl_xor(send.y, 16, 64);
        const uint4 o = odd ? make_uint4(recv.x, recv.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, recv.x, recv.y);
        const int start = odd ? i1 * 16 + kq * 4 - 4 : i0 * 16 + kq * 4;
        if (tv && !(p.debug & 1)) *reinterpret_cast<uint4*>(yrow + start) = o;
      }
    }
  }
#undef W12ROW
#undef GLOAD_PANELS
#undef L12IDX
#undef LSTORE_PANELS
}

bool ttvk_mlp_fused_supported(int dtype, int width, int inner) { return dtype == TTV_BF16 && width == 256 && inner % 32 == 0 && inner > 0; }

int ttvk_mlp_fused(const void* x, int ldx, const void* w12_folded, const void* w3_perm, int inner, void* y, int ldy,
                   const float* post_gain, float alpha, float eps, int M, hipStream_t s) {
  if (M == 0) return TTV_OK;
  TTV_CHECK_ARG(x && w12_folded && w3_perm && y, "mlp_fused: null buffer");
  TTV_CHECK_ARG(inner % 32 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "mlp_fused: inner %% 32, leading dims %% 8");
  TTV_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w12_folded % 16 == 0) && ((uintptr_t)w3_perm % 16 == 0) && ((uintptr_t)y % 16 == 0) && (inner * 2) % 16 == 0, "mlp_fused: 16-byte alignment");
  MlpDev d;
  d.x = (const bf16_t*)x; d.ldx = ldx; d.w12 = (const bf16_t*)w12_folded; d.w3p = (const bf16_t*)w3_perm; d.I = inner;
  d.y = (bf16_t*)y; d.ldy = ldy; d.post_gain = post_gain; d.alpha = alpha; d.eps = eps; d.M = M; d.debug = g_ttv_debug;
  // tokens per wave (NT*16): pick the tile size that needs the fewest full rounds of the 256 CUs, weighted by tile cost
  const int cus = 256;
  const long c2 = (long)ttv_cdiv(ttv_cdiv(M, 128), cus) * 2, c3 = (long)ttv_cdiv(ttv_cdiv(M, 192), cus) * 3;
  const int nt = (c3 < c2 && !(g_ttv_debug & 8)) ? 3 : 2;   // debug bit3 forces the 2-tile variant
  d.n_tiles = ttv_cdiv(M, 64 * nt);
  const int grid = d.n_tiles < cus ? d.n_tiles : cus;
  const size_t smem = (2 * 64 * 32 + 2 * 4 * 256) * sizeof(uint4);   // 96 KiB
  TtvProfScope prof(TTV_KC_GEMM_GEGLU, s);
#define LAUNCH_MLP(NT_, KEEL_)                                                                                      \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)k_mlp256<NT_, KEEL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((k_mlp256<NT_, KEEL_>), dim3(grid), dim3(256), smem, s, d);                                  \
  } while (0)
  if (post_gain) { if (nt == 3) LAUNCH_MLP(3, true); else LAUNCH_MLP(2, true); }
  else { if (nt == 3) LAUNCH_MLP(3, false); else LAUNCH_MLP(2, false); }
#undef LAUNCH_MLP
  TTV_CHECK_LAUNCH("mlp_fused");
  return TTV_OK;
}

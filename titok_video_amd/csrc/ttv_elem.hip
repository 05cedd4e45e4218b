// HBM-bound row kernels of the TiTok-Video path: RMSNorm, token/patch row initialisation, the encoder tail
// (ln_post -> proj_out -> FSQ), FSQ, patch gather/scatter, rotary apply, codebook histogram.
// One wave (64 lanes) owns one row; row statistics use xor-butterfly wave reductions; all loads are
// 4-element vectors (8 B bf16 / 16 B fp32) per lane, consecutive lanes on consecutive addresses.
#include "ttv_common.h"
#include "ttv_kernels.h"

#define ROWS_PER_BLOCK 4
#define MAX_ITERS 4  // width <= 64 lanes * 4 elems * 4 iters = 1024

// One lane's four values of a 32-element MX block (8 consecutive lanes): block maximum by three xor shuffles, E8M0 byte of the smallest
// power of two >= max / 448, the four e4m3 bytes.  Bit-identical to k_quant_mx_fp8 (no row factor) on the same values.
__device__ __forceinline__ int mx_quant4(const f32x4 y, int& byte) {
  float a = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
  a = fmaxf(a, __shfl_xor(a, 1));
  a = fmaxf(a, __shfl_xor(a, 2));
  a = fmaxf(a, __shfl_xor(a, 4));
  const uint32_t tb = __float_as_uint(a * (1.0f / 448.0f));
  byte = (int)((tb >> 23) & 0xFF) + ((tb & 0x7FFFFF) ? 1 : 0);
  byte = a > 0.f ? (byte < 1 ? 1 : (byte > 254 ? 254 : byte)) : 127;
  const float inv_blk = __uint_as_float((uint32_t)(254 - byte) << 23);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * inv_blk, y[1] * inv_blk, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * inv_blk, y[3] * inv_blk, w, true);
  return w;
}

// ------------------------------------------------------------------------------------------------
// RMSNorm with optional row gather/scatter maps.
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_rmsnorm(const TI* __restrict__ in, int ld_in, const int* __restrict__ src_rows,
                                                 TO* __restrict__ out, int ld_out, const int* __restrict__ dst_rows,
                                                 const float* __restrict__ gain, int rows, int d, float eps,
                                                 float* __restrict__ next_rstd, uint8_t* __restrict__ mx_q = nullptr,
                                                 uint8_t* __restrict__ mx_s = nullptr, int mx_ld = 0, int mx_nkp = 0, int split_image = 0) {
  // mx_q / mx_s (round 4, config #5): the row as STORED, additionally as block-scaled e4m3 (k_quant_mx_fp8's output for that row,
  // bit for bit) - the next linear's fp8 operand without a quantisation pass of its own.  Needs d % 128 == 0.
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  const int sr = src_rows ? src_rows[r] : r;
  const int dr = dst_rows ? dst_rows[r] : r;
  const TI* p = in + (size_t)sr * ld_in;
  f32x4 v[MAX_ITERS];
  float ss = 0.f;
  float so = 0.f;       // next_rstd: sum of squares of the row as STORED (rounded to TO) - what the next pre-norm will read
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < d) {
      v[it] = Vec4<TI>::load(p + c);
      ss += v[it][0] * v[it][0] + v[it][1] * v[it][1] + v[it][2] * v[it][2] + v[it][3] * v[it][3];
    }
  }
  ss = wave_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
  TO* q = out + (size_t)dr * ld_out;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < d) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
      f32x4 o = {v[it][0] * rstd * g[0], v[it][1] * rstd * g[1], v[it][2] * rstd * g[2], v[it][3] * rstd * g[3]};
      if (sizeof(TO) == 4 && split_image) {      // split-bf16 towers: the row as the next linear's split image (same 16 bytes per four elements)
        bf16x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[e] = (bf16_t)o[e]; lv[e] = (bf16_t)(o[e] - (float)hv[e]); }
        const uint2 hp = __builtin_bit_cast(uint2, hv), lp = __builtin_bit_cast(uint2, lv);
        *reinterpret_cast<uint4*>(q + c) = make_uint4(hp.x, hp.y, lp.x, lp.y);
      } else {
        Vec4<TO>::store(q + c, o);
      }
      if (next_rstd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float w = round_to<TO>(o[e]); so = fmaf(w, w, so); }
      }
      if (mx_q) {        // d % 128 == 0: lanes leave in 32-lane halves, never inside an 8-lane block
        const f32x4 yr = {round_to<TO>(o[0]), round_to<TO>(o[1]), round_to<TO>(o[2]), round_to<TO>(o[3])};
        int byte;
        const int w8 = mx_quant4(yr, byte);
        *reinterpret_cast<int*>(mx_q + (size_t)dr * d + c) = w8;
        if ((lane & 7) == 0) {
          const int b = c >> 5;
          mx_s[(size_t)dr * mx_ld + (b & 3) * mx_nkp + (b >> 2)] = (uint8_t)byte;
        }
      }
    }
  }
  if (next_rstd) {       // wave-uniform
    so = wave_sum(so);
    if (lane == 0) next_rstd[dr] = 1.0f / sqrtf(so / (float)d + eps);
  }
}

// rstd[r] = rsqrt(mean(x_r^2) + eps): the row statistic of an RMSNorm whose gain is folded into the next linear's weight and
// whose scale is applied to that GEMM's output rows (generic-width towers; the width-256 GEMM takes it from the register-resident row)
template <typename T>
__global__ __launch_bounds__(256) void k_row_rstd(const T* __restrict__ in, int ld_in, float* __restrict__ rstd, int rows, int d, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  const T* p = in + (size_t)r * ld_in;
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < d) {
      const f32x4 v = Vec4<T>::load(p + c);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) rstd[r] = 1.0f / sqrtf(ss / (float)d + eps);
}

// rows gathered / scattered through optional row maps, 16 bytes per thread and step (the encoder's last layer: latent rows -> compact)
__global__ __launch_bounds__(256) void k_copy_rows(const char* __restrict__ src, long ld_src, const int* __restrict__ src_rows, char* __restrict__ dst,
                                                   long ld_dst, const int* __restrict__ dst_rows, int rows, int chunks) {
  const long total = (long)rows * chunks;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    const uint4 v = *reinterpret_cast<const uint4*>(src + (long)(src_rows ? src_rows[r] : r) * ld_src + c * 16);
    *reinterpret_cast<uint4*>(dst + (long)(dst_rows ? dst_rows[r] : r) * ld_dst + c * 16) = v;
  }
}
int ttvk_copy_rows(const void* src, int64_t ld_src_bytes, const int* src_rows, void* dst, int64_t ld_dst_bytes, const int* dst_rows, int rows,
                   int row_bytes, hipStream_t s) {
  if (rows == 0 || row_bytes == 0) return TTV_OK;
  TTV_CHECK_ARG(src && dst && row_bytes % 16 == 0 && ld_src_bytes % 16 == 0 && ld_dst_bytes % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                "copy_rows: 16-byte rows");
  const long total = (long)rows * (row_bytes / 16);
  int blocks = (int)((total + 255) / 256);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(k_copy_rows, dim3(blocks), dim3(256), 0, s, (const char*)src, (long)ld_src_bytes, src_rows, (char*)dst, (long)ld_dst_bytes, dst_rows, rows, row_bytes / 16);
  TTV_CHECK_LAUNCH("copy_rows");
  return TTV_OK;
}

int ttvk_row_rstd(const void* in, int dtype, int ld_in, float* rstd, int rows, int d, float eps, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(d % 4 == 0 && d <= 64 * 4 * MAX_ITERS && ld_in % 4 == 0, "row_rstd: width / leading dim");
  dim3 grid(ttv_cdiv(rows, ROWS_PER_BLOCK));
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_row_rstd<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, ld_in, rstd, rows, d, eps);
  else hipLaunchKernelGGL((k_row_rstd<float>), grid, dim3(256), 0, s, (const float*)in, ld_in, rstd, rows, d, eps);
  TTV_CHECK_LAUNCH("row_rstd");
  return TTV_OK;
}

template <typename TI, typename TO>
static int launch_rmsnorm(const void* in, int ld_in, const int* src_rows, void* out, int ld_out, const int* dst_rows,
                          const float* gain, int rows, int d, float eps, hipStream_t s, float* next_rstd, void* mx_q, void* mx_s, int split_image) {
  if (rows == 0) return TTV_OK;
  TtvProfScope prof(TTV_KC_RMSNORM, s);
  const int nkp = (d / 128 + 3) / 4 * 4;
  hipLaunchKernelGGL((k_rmsnorm<TI, TO>), dim3(ttv_cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0, s, (const TI*)in, ld_in,
                     src_rows, (TO*)out, ld_out, dst_rows, gain, rows, d, eps, next_rstd, (uint8_t*)mx_q, (uint8_t*)mx_s, 4 * nkp, nkp, split_image);
  TTV_CHECK_LAUNCH("rmsnorm");
  return TTV_OK;
}

// Width 256, bf16 in and out, no row maps and no side outputs - the 31 norms of a training step's tape forward (and every other plain
// call at that width): half a wave per row with 16-byte accesses, eight rows of a wave in flight at once, 1 152 blocks for 36 864 rows.
// The one-row-per-wave kernel above issues ONE 8-byte load per lane and waits for it: 11 us for 38 MB (3.5 TB/s) - latency, not bandwidth.
__global__ __launch_bounds__(256) void k_rmsnorm256_bf16(const bf16_t* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int ld_out,
                                                         const float* __restrict__ gain, int rows, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, c = (lane & 31) * 8;
  const int r0 = (blockIdx.x * 4 + wave) * 8;
  if (r0 >= rows) return;
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gain + c), g1 = *reinterpret_cast<const f32x4*>(gain + c + 4);
  bf16x8 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + 2 * k + half;
    v[k] = *reinterpret_cast<const bf16x8*>(in + (size_t)(r < rows ? r : rows - 1) * ld_in + c);
  }
  float ss[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ss[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)v[k][e]; ss[k] = fmaf(f, f, ss[k]); }
  }
  // 32-lane sums, partners ^ 1, 2, 4, 8, 16 in that order, every exchange on the vector ALU (ttv_common.h)
#pragma unroll
  for (int k = 0; k < 4; ++k) ss[k] += wave_xor_dpp1(ss[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) ss[k] += wave_xor_dpp2(ss[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) ss[k] += wave_xor_dpp4(ss[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) ss[k] += wave_xor_dpp8(ss[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned u = __float_as_uint(ss[k]);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    ss[k] = __uint_as_float(a0) + __uint_as_float(a1);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + 2 * k + half;
    const float rstd = 1.0f / sqrtf(ss[k] * (1.0f / 256.0f) + eps);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)((float)v[k][e] * rstd * g0[e]); o[e + 4] = (bf16_t)((float)v[k][e + 4] * rstd * g1[e]); }
    if (r < rows) *reinterpret_cast<bf16x8*>(out + (size_t)r * ld_out + c) = o;
  }
}

int ttvk_rmsnorm(const void* in, int in_dtype, int ld_in, const int* src_rows, void* out, int out_dtype, int ld_out,
                 const int* dst_rows, const float* gain, int rows, int d, float eps, hipStream_t s, float* next_rstd, void* mx_q, void* mx_s,
                 int split_image) {
  TTV_CHECK_ARG(d % 4 == 0 && d <= 64 * 4 * MAX_ITERS, "rmsnorm: width %d must be a multiple of 4 and <= 1024", d);
  TTV_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0, "rmsnorm: leading dims must be multiples of 4");
  TTV_CHECK_ARG(!mx_q || (mx_s && d % 128 == 0 && (uintptr_t)mx_q % 4 == 0), "rmsnorm: the block-scaled fp8 side output needs its scale buffer and width %% 128 == 0");
  static const bool fast256 = !(getenv("TTV_RMSNORM256") && getenv("TTV_RMSNORM256")[0] == '0');      // A/B
  if (fast256 && rows > 0 && d == 256 && in_dtype == TTV_BF16 && out_dtype == TTV_BF16 && !src_rows && !dst_rows && !next_rstd && !mx_q && !split_image &&
      ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)gain % 16 == 0)) {
    hipLaunchKernelGGL(k_rmsnorm256_bf16, dim3(ttv_cdiv(rows, 32)), dim3(256), 0, s, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, gain, rows, eps);
    TTV_CHECK_LAUNCH("rmsnorm256");
    return TTV_OK;
  }
  if (in_dtype == TTV_F32 && out_dtype == TTV_F32) return launch_rmsnorm<float, float>(in, ld_in, src_rows, out, ld_out, dst_rows, gain, rows, d, eps, s, next_rstd, mx_q, mx_s, split_image);
  if (in_dtype == TTV_F32 && out_dtype == TTV_BF16) return launch_rmsnorm<float, bf16_t>(in, ld_in, src_rows, out, ld_out, dst_rows, gain, rows, d, eps, s, next_rstd, mx_q, mx_s, split_image);
  if (in_dtype == TTV_BF16 && out_dtype == TTV_BF16) return launch_rmsnorm<bf16_t, bf16_t>(in, ld_in, src_rows, out, ld_out, dst_rows, gain, rows, d, eps, s, next_rstd, mx_q, mx_s, split_image);
  if (in_dtype == TTV_BF16 && out_dtype == TTV_F32) return launch_rmsnorm<bf16_t, float>(in, ld_in, src_rows, out, ld_out, dst_rows, gain, rows, d, eps, s, next_rstd, mx_q, mx_s, split_image);
  ttv_set_error("rmsnorm: bad dtypes %d %d", in_dtype, out_dtype);
  return TTV_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------
// Row-wise fp8 (OCP e4m3) quantisation for the mixed bf16 / fp8 linears (BASELINE config #5; not a reference feature - the
// reference runs bf16 autocast, configs/tiny.yaml:70): optionally the RMSNorm in front of the linear (transformer.py:48,86) in
// the same pass.  y = gain ? x * rsqrt(mean(x^2) + eps) * gain : x;  scale[r] = max|y| / 448;  q = round_e4m3(y / scale[r]).
// One wave per row, 4 elements per lane and step; v_cvt_pk_fp8_f32 packs two values per call.
// ------------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void k_quant_rows_fp8(const TI* __restrict__ in, int ld_in, const float* __restrict__ gain, float eps,
                                                        uint8_t* __restrict__ out, int ld_out, float* __restrict__ scales, int rows, int d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  const TI* p = in + (size_t)r * ld_in;
  f32x4 v[MAX_ITERS];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      v[it] = Vec4<TI>::load(p + c);
      ss += v[it][0] * v[it][0] + v[it][1] * v[it][1] + v[it][2] * v[it][2] + v[it][3] * v[it][3];
    }
  }
  float amax = 0.f;
  if (gain) {
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
#pragma unroll
    for (int it = 0; it < MAX_ITERS; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < d) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[it][e] = round_to<TI>(v[it][e] * rstd * g[e]);     // what the bf16 path would hand the linear
      }
    }
  }
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[it][e]));
  amax = wave_max(amax);
  const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / scale;
  if (lane == 0) scales[r] = scale;
  uint8_t* q = out + (size_t)r * ld_out;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < d) {
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[it][0] * inv, v[it][1] * inv, w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[it][2] * inv, v[it][3] * inv, w, true);
      *reinterpret_cast<int*>(q + c) = w;
    }
  }
}

int ttvk_quant_rows_fp8(const void* in, int in_dtype, int ld_in, const float* gain, float eps, void* out, int ld_out, float* scales, int rows,
                        int d, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(d % 4 == 0 && d <= 64 * 4 * MAX_ITERS, "quant_rows_fp8: width %d must be a multiple of 4 and <= 1024", d);
  TTV_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0 && (uintptr_t)out % 4 == 0, "quant_rows_fp8: leading dims must be multiples of 4");
  dim3 grid(ttv_cdiv(rows, ROWS_PER_BLOCK));
  if (in_dtype == TTV_BF16) hipLaunchKernelGGL((k_quant_rows_fp8<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, ld_in, gain, eps, (uint8_t*)out, ld_out, scales, rows, d);
  else if (in_dtype == TTV_F32) hipLaunchKernelGGL((k_quant_rows_fp8<float>), grid, dim3(256), 0, s, (const float*)in, ld_in, gain, eps, (uint8_t*)out, ld_out, scales, rows, d);
  else { ttv_set_error("quant_rows_fp8: bad dtype"); return TTV_ERR_INVALID; }
  TTV_CHECK_LAUNCH("quant_rows_fp8");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Split image of an fp32 weight matrix for the three-pass bf16 linears (k_gemm_f32<.., SPLIT>): per aligned group of four k values the
// 16 bytes (hi0..3 | lo0..3), hi = bf16(x) (round to nearest even), lo = bf16(x - hi).  One thread per group.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split3_pack(const float* __restrict__ w, int ldw, uint4* __restrict__ out, int ldo4, int rows, int k4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * k4) return;
  const int r = (int)(i / k4), c = (int)(i % k4);
  const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)r * ldw + c * 4);
  bf16x4 hv, lv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hv[e] = (bf16_t)v[e];
    lv[e] = (bf16_t)(v[e] - (float)hv[e]);
  }
  const uint2 hp = __builtin_bit_cast(uint2, hv), lp = __builtin_bit_cast(uint2, lv);
  out[(size_t)r * ldo4 + c] = make_uint4(hp.x, hp.y, lp.x, lp.y);
}

int ttvk_split3_pack(const float* w, int ldw, void* out, int ldo, int rows, int K, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(K > 0 && K % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)out % 16 == 0, "split3_pack: K and leading dims must be multiples of 4, pointers 16-byte aligned");
  const long n = (long)rows * (K / 4);
  hipLaunchKernelGGL(k_split3_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, ldw, (uint4*)out, ldo / 4, rows, K / 4);
  TTV_CHECK_LAUNCH("split3_pack");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// MX (OCP microscaling) e4m3 quantisation for the block-scaled fp8 linears (BASELINE config #5): every 32 consecutive elements of a
// row share one E8M0 scale 2^E, E = ceil(log2(max|block| / 448)) (the smallest power of two that brings the block into e4m3's range);
// q = round_e4m3(y / 2^E).  The scale bytes are what v_mfma_scale_f32_16x16x128_f8f6f4 takes as its per-lane scale operand
// (k_gemm_fp8_dma<.., MX>).  Optional per-row fp32 factor (weights: row_scales[r] = max|row| / 448, blocks then scale the row-normalised
// values, E <= 0); activations pass NULL - their per-row factor is the rstd of the folded pre-norm, applied by the GEMM epilogue.
// Scale layout (this library's, chosen for the GEMM's lanes): block b = k / 32 of row r lives at mx[r * ld_mx + (b & 3) * nkp + (b >> 2)],
// nkp = round_up(d / 128, 4), ld_mx = 4 * nkp: lane group kq = b & 3 of the MFMA finds the bytes of four consecutive 128-element
// k-tiles in one aligned dword.  One wave per row, 4 elements per lane and step (a block = 8 lanes: three DPP-free xor shuffles).
// ------------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void k_quant_mx_fp8(const TI* __restrict__ in, int ld_in, uint8_t* __restrict__ out, int ld_out,
                                                      uint8_t* __restrict__ mx, int ld_mx, int nkp, float* __restrict__ row_scales,
                                                      int rows, int d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  const TI* p = in + (size_t)r * ld_in;
  float inv_row = 1.0f;
  if (row_scales) {                                 // weights (packed once): a first pass for the row maximum, the row is re-read from the cache
    float amax = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 y = Vec4<TI>::load(p + c);
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
    }
    amax = wave_max(amax);
    const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    inv_row = 1.0f / scale;
    if (lane == 0) row_scales[r] = scale;
  }
  uint8_t* q = out + (size_t)r * ld_out;
  uint8_t* m = mx + (size_t)r * ld_mx;
  // d is a multiple of 128: in the last step lanes 0-31 may work while 32-63 idle, never a part of an 8-lane block
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 y = Vec4<TI>::load(p + c) * inv_row;
    float a = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
    a = fmaxf(a, __shfl_xor(a, 1));
    a = fmaxf(a, __shfl_xor(a, 2));
    a = fmaxf(a, __shfl_xor(a, 4));
    // E8M0 byte = biased exponent of the smallest power of two >= a / 448
    const uint32_t tb = __float_as_uint(a * (1.0f / 448.0f));
    int byte = (int)((tb >> 23) & 0xFF) + ((tb & 0x7FFFFF) ? 1 : 0);
    byte = a > 0.f ? (byte < 1 ? 1 : (byte > 254 ? 254 : byte)) : 127;
    const float inv_blk = __uint_as_float((uint32_t)(254 - byte) << 23);      // 2^-(byte - 127)
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * inv_blk, y[1] * inv_blk, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * inv_blk, y[3] * inv_blk, w, true);
    *reinterpret_cast<int*>(q + c) = w;
    if ((lane & 7) == 0) {
      const int b = c >> 5;
      m[(b & 3) * nkp + (b >> 2)] = (uint8_t)byte;
    }
  }
}

int64_t ttvk_mx_scale_ld(int d) { return 4 * (int64_t)((d / 128 + 3) / 4 * 4); }

int ttvk_quant_mx_fp8(const void* in, int in_dtype, int ld_in, void* out, int ld_out, void* mx, float* row_scales, int rows, int d, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(d > 0 && d % 128 == 0, "quant_mx_fp8: width %d must be a multiple of 128", d);
  TTV_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0 && (uintptr_t)out % 4 == 0 && (uintptr_t)mx % 4 == 0, "quant_mx_fp8: alignment");
  const int nkp = (d / 128 + 3) / 4 * 4;
  dim3 grid(ttv_cdiv(rows, ROWS_PER_BLOCK));
  if (in_dtype == TTV_BF16) hipLaunchKernelGGL((k_quant_mx_fp8<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, ld_in, (uint8_t*)out, ld_out, (uint8_t*)mx, 4 * nkp, nkp, row_scales, rows, d);
  else if (in_dtype == TTV_F32) hipLaunchKernelGGL((k_quant_mx_fp8<float>), grid, dim3(256), 0, s, (const float*)in, ld_in, (uint8_t*)out, ld_out, (uint8_t*)mx, 4 * nkp, nkp, row_scales, rows, d);
  else { ttv_set_error("quant_mx_fp8: bad dtype"); return TTV_ERR_INVALID; }
  TTV_CHECK_LAUNCH("quant_mx_fp8");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Rows that hold RMSNorm(mask_token * ones(d)) * gain: encoder latent rows (blocks.py:96), decoder patch rows
// (blocks.py:167).  Every such row is the same vector; mean(m^2) over a constant row is m^2.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fill_const_rows(T* __restrict__ x, int ld, const int* __restrict__ rows_map, int rows,
                                                         int d, const float* __restrict__ mask_token,
                                                         const float* __restrict__ gain, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  const float m = round_to<T>(mask_token[0]);  // mask_token.to(dtype)
  const float rstd = 1.0f / sqrtf(m * m + eps);
  T* q = x + (size_t)rows_map[r] * ld;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
    f32x4 o = {m * rstd * g[0], m * rstd * g[1], m * rstd * g[2], m * rstd * g[3]};
    Vec4<T>::store(q + c, o);
  }
}

int ttvk_fill_const_rows(void* x, int dtype, int ld, const int* rows_map, int rows, int d, const float* mask_token,
                         const float* gain, float eps, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(d % 4 == 0, "fill_const_rows: width %% 4");
  dim3 grid(ttv_cdiv(rows, ROWS_PER_BLOCK));
  if (dtype == TTV_BF16)
    hipLaunchKernelGGL((k_fill_const_rows<bf16_t>), grid, dim3(256), 0, s, (bf16_t*)x, ld, rows_map, rows, d, mask_token, gain, eps);
  else
    hipLaunchKernelGGL((k_fill_const_rows<float>), grid, dim3(256), 0, s, (float*)x, ld, rows_map, rows, d, mask_token, gain, eps);
  TTV_CHECK_LAUNCH("fill_const_rows");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Decoder latent rows: RMSNorm_t(proj_in(codes) + bias + mask_token) (blocks.py:125,166).  token_size <= TTV_MAX_FSQ keeps the code
// row in registers; wider tokens (<= TTV_MAX_TOKEN, the L2 quantiser's) re-read it from the cache in the feature loop.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_dec_embed(const T* __restrict__ codes, int C, const T* __restrict__ w /*[d,C]*/,
                                                   const T* __restrict__ bias, const float* __restrict__ mask_token,
                                                   const float* __restrict__ gain, T* __restrict__ x, int ld,
                                                   const int* __restrict__ rows_map, int rows, int d, float eps, T* __restrict__ hpre) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  float cz[TTV_MAX_FSQ];
#pragma unroll
  for (int c = 0; c < TTV_MAX_FSQ; ++c) cz[c] = c < C ? Cvt<T>::to_f(codes[(size_t)r * C + c]) : 0.f;
  const T* crow = codes + (size_t)r * C;
  const float m = round_to<T>(mask_token[0]);
  float v[MAX_ITERS][4];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = (it * 64 + lane) * 4 + e;
      float h = 0.f;
      if (f < d) {
        if (C <= TTV_MAX_FSQ) {
#pragma unroll
          for (int c = 0; c < TTV_MAX_FSQ; ++c)
            if (c < C) h += cz[c] * Cvt<T>::to_f(w[(size_t)f * C + c]);
        } else if ((C & 3) == 0) {      // wide tokens (the L2 quantiser's 32 / 64): four elements per load, same summation order
          const T* wrow = w + (size_t)f * C;
          for (int c = 0; c < C; c += 4) {
            const f32x4 cv = Vec4<T>::load(crow + c), wv = Vec4<T>::load(wrow + c);
            h += cv[0] * wv[0];
            h += cv[1] * wv[1];
            h += cv[2] * wv[2];
            h += cv[3] * wv[3];
          }
        } else {
          for (int c = 0; c < C; ++c) h += Cvt<T>::to_f(crow[c]) * Cvt<T>::to_f(w[(size_t)f * C + c]);
        }
        h = round_to<T>(h + Cvt<T>::to_f(bias[f]));  // nn.Linear output in dtype
        h = round_to<T>(h + m);                      // + mask_token.to(dtype)
      }
      v[it][e] = h;
      ss += h * h;
    }
    if (hpre && (it * 64 + lane) * 4 < d)   // training tape: the pre-norm row proj_in(codes) + bias + mask_token
      Vec4<T>::store(hpre + (size_t)r * d + (it * 64 + lane) * 4, (f32x4){v[it][0], v[it][1], v[it][2], v[it][3]});
  }
  ss = wave_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
  T* q = x + (size_t)rows_map[r] * ld;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c0 = (it * 64 + lane) * 4;
    if (c0 < d) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c0);
      f32x4 o = {v[it][0] * rstd * g[0], v[it][1] * rstd * g[1], v[it][2] * rstd * g[2], v[it][3] * rstd * g[3]};
      Vec4<T>::store(q + c0, o);
    }
  }
}

int ttvk_dec_embed(const void* codes, int C, const void* w, const void* bias, const float* mask_token, const float* gain,
                   void* x, int dtype, int ld, const int* rows_map, int rows, int d, float eps, hipStream_t s) {
  return ttvk_dec_embed_ex(codes, C, w, bias, mask_token, gain, x, dtype, ld, rows_map, rows, d, eps, nullptr, s);
}

int ttvk_dec_embed_ex(const void* codes, int C, const void* w, const void* bias, const float* mask_token, const float* gain, void* x,
                      int dtype, int ld, const int* rows_map, int rows, int d, float eps, void* hpre, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(C >= 1 && C <= TTV_MAX_TOKEN, "dec_embed: token_size %d out of range", C);
  TTV_CHECK_ARG(d % 4 == 0 && d <= 1024, "dec_embed: width");
  dim3 grid(ttv_cdiv(rows, ROWS_PER_BLOCK));
  if (dtype == TTV_BF16)
    hipLaunchKernelGGL((k_dec_embed<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)codes, C, (const bf16_t*)w, (const bf16_t*)bias, mask_token, gain, (bf16_t*)x, ld, rows_map, rows, d, eps, (bf16_t*)hpre);
  else
    hipLaunchKernelGGL((k_dec_embed<float>), grid, dim3(256), 0, s, (const float*)codes, C, (const float*)w, (const float*)bias, mask_token, gain, (float*)x, ld, rows_map, rows, d, eps, (float*)hpre);
  TTV_CHECK_LAUNCH("dec_embed");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// FSQ (fsq.py:78-135).  The arithmetic is written op by op (no FMA contraction) so that it follows the
// reference's sequence of fp32 tensor ops: tanh(z+shift)*half_l-offset -> round-half-even -> /half_width ->
// *half_width+half_width -> *basis -> sum -> int32 truncation.
// ------------------------------------------------------------------------------------------------
struct FsqDev {
  int n;
  float half_l[TTV_MAX_FSQ], offset[TTV_MAX_FSQ], shift[TTV_MAX_FSQ], half_width[TTV_MAX_FSQ], basis[TTV_MAX_FSQ];
  int levels[TTV_MAX_FSQ], ibasis[TTV_MAX_FSQ];
};

static FsqDev to_dev(const ttv_fsq_params* p) {
  FsqDev d;
  d.n = p->n;
  for (int i = 0; i < TTV_MAX_FSQ; ++i) {
    d.half_l[i] = p->half_l[i]; d.offset[i] = p->offset[i]; d.shift[i] = p->shift[i];
    d.half_width[i] = p->half_width[i]; d.basis[i] = (float)p->basis[i];
    d.levels[i] = p->levels[i]; d.ibasis[i] = p->basis[i];
  }
  return d;
}

__device__ __forceinline__ float fsq_channel(const FsqDev& p, int c, float z, float* bounded_out, float* code_out) {
  const float b = __fsub_rn(__fmul_rn(tanhf(__fadd_rn(z, p.shift[c])), p.half_l[c]), p.offset[c]);
  const float q = rintf(b);                       // round half to even, like torch.round
  const float code = __fdiv_rn(q, p.half_width[c]);
  const float zhat = __fadd_rn(__fmul_rn(code, p.half_width[c]), p.half_width[c]);
  *bounded_out = b;
  *code_out = code;
  return __fmul_rn(zhat, p.basis[c]);
}

template <typename TZ, typename TC>
__global__ __launch_bounds__(256) void k_fsq_forward(FsqDev p, const TZ* __restrict__ z, int rows, TC* __restrict__ codes,
                                                     int* __restrict__ indices, float* __restrict__ bounded) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float acc = 0.f;
  for (int c = 0; c < p.n; ++c) {
    float b, code;
    acc = __fadd_rn(acc, fsq_channel(p, c, Cvt<TZ>::to_f(z[(size_t)r * p.n + c]), &b, &code));
    codes[(size_t)r * p.n + c] = Cvt<TC>::from_f(code);
    if (bounded) bounded[(size_t)r * p.n + c] = b;
  }
  indices[r] = (int)acc;
}

int ttvk_fsq_forward(const ttv_fsq_params* p, const void* z, int z_dtype, int rows, void* codes, int codes_dtype, int* indices,
                     float* bounded, hipStream_t s) {
  TTV_CHECK_ARG(p && p->n >= 1 && p->n <= TTV_MAX_FSQ, "fsq: codebook_dim out of range");
  if (rows == 0) return TTV_OK;
  FsqDev d = to_dev(p);
  dim3 grid(ttv_cdiv(rows, 256));
  if (z_dtype == TTV_F32 && codes_dtype == TTV_F32)
    hipLaunchKernelGGL((k_fsq_forward<float, float>), grid, dim3(256), 0, s, d, (const float*)z, rows, (float*)codes, indices, bounded);
  else if (z_dtype == TTV_F32 && codes_dtype == TTV_BF16)
    hipLaunchKernelGGL((k_fsq_forward<float, bf16_t>), grid, dim3(256), 0, s, d, (const float*)z, rows, (bf16_t*)codes, indices, bounded);
  else if (z_dtype == TTV_BF16 && codes_dtype == TTV_BF16)
    hipLaunchKernelGGL((k_fsq_forward<bf16_t, bf16_t>), grid, dim3(256), 0, s, d, (const bf16_t*)z, rows, (bf16_t*)codes, indices, bounded);
  else if (z_dtype == TTV_BF16 && codes_dtype == TTV_F32)
    hipLaunchKernelGGL((k_fsq_forward<bf16_t, float>), grid, dim3(256), 0, s, d, (const bf16_t*)z, rows, (float*)codes, indices, bounded);
  else { ttv_set_error("fsq: bad dtypes"); return TTV_ERR_INVALID; }
  TTV_CHECK_LAUNCH("fsq_forward");
  return TTV_OK;
}

template <typename TC>
__global__ __launch_bounds__(256) void k_fsq_indices_to_codes(FsqDev p, const int* __restrict__ indices, int rows,
                                                              TC* __restrict__ codes) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int idx = indices[r];
  for (int c = 0; c < p.n; ++c) {
    const int lvl = (idx / p.ibasis[c]) % p.levels[c];            // fsq.py:111-115
    const float hw = p.half_width[c];
    codes[(size_t)r * p.n + c] = Cvt<TC>::from_f(__fdiv_rn(__fsub_rn((float)lvl, hw), hw));  // fsq.py:96-98
  }
}

int ttvk_fsq_indices_to_codes(const ttv_fsq_params* p, const int* indices, int rows, void* codes, int codes_dtype, hipStream_t s) {
  TTV_CHECK_ARG(p && p->n >= 1 && p->n <= TTV_MAX_FSQ, "fsq: codebook_dim out of range");
  if (rows == 0) return TTV_OK;
  FsqDev d = to_dev(p);
  dim3 grid(ttv_cdiv(rows, 256));
  if (codes_dtype == TTV_F32) hipLaunchKernelGGL((k_fsq_indices_to_codes<float>), grid, dim3(256), 0, s, d, indices, rows, (float*)codes);
  else hipLaunchKernelGGL((k_fsq_indices_to_codes<bf16_t>), grid, dim3(256), 0, s, d, indices, rows, (bf16_t*)codes);
  TTV_CHECK_LAUNCH("fsq_indices_to_codes");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Encoder tail (blocks.py:101-103 + fsq.py:123-135): gather latent rows -> ln_post -> proj_out (d -> C) + bias -> FSQ.
// One wave per latent token; the C <= 8 dot products are wave reductions; z stays fp32 into the quantiser.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_enc_tail(const T* __restrict__ x, int ld, const int* __restrict__ rows_map, int rows,
                                                  int d, const float* __restrict__ gain, float eps,
                                                  const T* __restrict__ w /*[C,d]*/, const T* __restrict__ bias, int C,
                                                  FsqDev p, int have_fsq, float* __restrict__ z_out, T* __restrict__ codes,
                                                  int* __restrict__ indices, float* __restrict__ bounded) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (r >= rows) return;
  const T* px = x + (size_t)rows_map[r] * ld;
  f32x4 v[MAX_ITERS];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < d) {
      v[it] = Vec4<T>::load(px + c);
      ss += v[it][0] * v[it][0] + v[it][1] * v[it][1] + v[it][2] * v[it][2] + v[it][3] * v[it][3];
    } else {
      v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  ss = wave_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
#pragma unroll
  for (int it = 0; it < MAX_ITERS; ++it) {
    const int c = (it * 64 + lane) * 4;
    if (c < d) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[it][e] = round_to<T>(v[it][e] * rstd * g[e]);  // ln_post output in dtype
    }
  }
  float zc[TTV_MAX_FSQ];
#pragma unroll
  for (int c = 0; c < TTV_MAX_FSQ; ++c) zc[c] = 0.f;
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_ITERS; ++it) {
      const int f = (it * 64 + lane) * 4;
      if (f < d) {
        const f32x4 wv = Vec4<T>::load(w + (size_t)c * d + f);
        acc += v[it][0] * wv[0] + v[it][1] * wv[1] + v[it][2] * wv[2] + v[it][3] * wv[3];
      }
    }
    const float zv = wave_sum(acc) + Cvt<T>::to_f(bias[c]);
    if (lane == 0 && z_out) z_out[(size_t)r * C + c] = zv;
    if (have_fsq) {                 // C <= TTV_MAX_FSQ (checked by the host): static register indices
#pragma unroll
      for (int k = 0; k < TTV_MAX_FSQ; ++k)
        if (k == c) zc[k] = zv;
    }
  }
  if (lane == 0 && have_fsq) {
    float idx = 0.f;
    for (int c = 0; c < C; ++c) {
      {
        float b, code;
        idx = __fadd_rn(idx, fsq_channel(p, c, zc[c], &b, &code));
        codes[(size_t)r * C + c] = Cvt<T>::from_f(code);
        if (bounded) bounded[(size_t)r * C + c] = b;
      }
    }
    if (have_fsq) indices[r] = (int)idx;
  }
}

int ttvk_enc_tail(const void* x, int dtype, int ld, const int* rows_map, int rows, int d, const float* gain, float eps,
                  const void* w, const void* bias, int C, const ttv_fsq_params* fsq, float* z_out, void* codes, int* indices,
                  float* bounded, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(C >= 1 && C <= (fsq ? TTV_MAX_FSQ : TTV_MAX_TOKEN), "enc_tail: token_size %d out of range", C);
  TTV_CHECK_ARG(d % 4 == 0 && d <= 1024, "enc_tail: width");
  TTV_CHECK_ARG(!fsq || fsq->n == C, "enc_tail: fsq dim != token_size");
  FsqDev p;
  if (fsq) p = to_dev(fsq); else { p = FsqDev(); p.n = 0; }
  dim3 grid(ttv_cdiv(rows, ROWS_PER_BLOCK));
  if (dtype == TTV_BF16)
    hipLaunchKernelGGL((k_enc_tail<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, ld, rows_map, rows, d, gain, eps, (const bf16_t*)w, (const bf16_t*)bias, C, p, fsq ? 1 : 0, z_out, (bf16_t*)codes, indices, bounded);
  else
    hipLaunchKernelGGL((k_enc_tail<float>), grid, dim3(256), 0, s, (const float*)x, ld, rows_map, rows, d, gain, eps, (const float*)w, (const float*)bias, C, p, fsq ? 1 : 0, z_out, (float*)codes, indices, bounded);
  TTV_CHECK_LAUNCH("enc_tail");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Patch gather / scatter (utils.py:26-51).  A segment = the pw contiguous pixels of one (c, ipt, iph) image row
// of a patch; the patch vector is stored in (c,pt,ph,pw) order so a segment is one 16-byte vector for pw = 8 bf16.
// ------------------------------------------------------------------------------------------------

template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void k_patch_copy(ClipPtrs clips, const int* __restrict__ clip_desc, int clip0, int pt, int ph,
                                                    int pw, int C, T* __restrict__ patches, int ld) {
  const int* ds = clip_desc + (size_t)(clip0 + blockIdx.y) * 8;
  const int Tn = ds[0], H = ds[1], W = ds[2], gh = ds[4], gw = ds[5], base = ds[6];
  const int P = ds[3] * gh * gw;
  const int nseg = C * pt * ph;
  const int total = P * nseg;
  T* clip = reinterpret_cast<T*>(clips.p[blockIdx.y]);
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int p = idx / nseg, sg = idx - p * nseg;
    const int c = sg / (pt * ph), ipt = (sg / ph) % pt, iph = sg % ph;
    const int gwi = p % gw, ghi = (p / gw) % gh, gti = p / (gw * gh);
    T* src = clip + (((size_t)c * Tn + gti * pt + ipt) * H + ghi * ph + iph) * W + gwi * pw;
    T* dst = patches + (size_t)(base + p) * ld + sg * pw;
    if ((pw * sizeof(T)) % 16 == 0) {
      for (int e = 0; e < (int)(pw * sizeof(T) / 16); ++e) {
        if (SCATTER) reinterpret_cast<uint4*>(src)[e] = reinterpret_cast<const uint4*>(dst)[e];
        else reinterpret_cast<uint4*>(dst)[e] = reinterpret_cast<const uint4*>(src)[e];
      }
    } else {
      for (int e = 0; e < pw; ++e) {
        if (SCATTER) src[e] = dst[e]; else dst[e] = src[e];
      }
    }
  }
}

int ttvk_patch_copy(bool scatter, void* const* clips, const int* clip_desc, int clip0, int n_clips, int pt, int ph, int pw,
                    int C, void* patches, int ld, int dtype, int max_patches, hipStream_t s) {
  TTV_CHECK_ARG(n_clips >= 0 && n_clips <= TTV_MAX_CLIPS_PER_LAUNCH, "patch_copy: at most %d clips per call", TTV_MAX_CLIPS_PER_LAUNCH);
  if (n_clips == 0 || max_patches == 0) return TTV_OK;
  ClipPtrs cp;
  for (int i = 0; i < n_clips; ++i) cp.p[i] = clips[i];
  TtvProfScope prof(TTV_KC_PATCH, s);
  const int total = max_patches * C * pt * ph;
  dim3 grid(ttv_cdiv(total, 256) > 4096 ? 4096 : ttv_cdiv(total, 256), n_clips);
  if (dtype == TTV_BF16) {
    if (scatter) hipLaunchKernelGGL((k_patch_copy<bf16_t, true>), grid, dim3(256), 0, s, cp, clip_desc, clip0, pt, ph, pw, C, (bf16_t*)patches, ld);
    else hipLaunchKernelGGL((k_patch_copy<bf16_t, false>), grid, dim3(256), 0, s, cp, clip_desc, clip0, pt, ph, pw, C, (bf16_t*)patches, ld);
  } else {
    if (scatter) hipLaunchKernelGGL((k_patch_copy<float, true>), grid, dim3(256), 0, s, cp, clip_desc, clip0, pt, ph, pw, C, (float*)patches, ld);
    else hipLaunchKernelGGL((k_patch_copy<float, false>), grid, dim3(256), 0, s, cp, clip_desc, clip0, pt, ph, pw, C, (float*)patches, ld);
  }
  TTV_CHECK_LAUNCH("patch_copy");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Rotary apply, standalone (rope.py:19-27).  The tower path fuses this into the QKV GEMM epilogue.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_rope_apply(T* __restrict__ x, int ld, int rows, int heads,
                                                    const float* __restrict__ cs, float sgn) {
  // thread -> (row, head, 4 consecutive dims = 2 complex pairs)
  const long total = (long)rows * heads * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q4 = (int)(i % 16), h = (int)((i / 16) % heads);
    const long r = i / (16 * heads);
    T* p = x + r * ld + h * 64 + q4 * 4;
    const float* c = cs + r * 64 + q4 * 2;
    f32x4 v = Vec4<T>::load(p);
    const float c0 = c[0], c1 = c[1], s0 = sgn * c[32], s1 = sgn * c[33];   // sgn = -1: inverse rotation (backward)
    f32x4 o = {v[0] * c0 - v[1] * s0, v[0] * s0 + v[1] * c0, v[2] * c1 - v[3] * s1, v[2] * s1 + v[3] * c1};
    Vec4<T>::store(p, o);
  }
}

int ttvk_rope_apply(void* x, int dtype, int ld, int rows, int heads, const float* cs, hipStream_t s) {
  return ttvk_rope_apply_dir(x, dtype, ld, rows, heads, cs, 0, s);
}

int ttvk_rope_apply_dir(void* x, int dtype, int ld, int rows, int heads, const float* cs, int conj, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  const float sgn = conj ? -1.f : 1.f;
  const long total = (long)rows * heads * 16;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_rope_apply<bf16_t>), dim3(blocks), dim3(256), 0, s, (bf16_t*)x, ld, rows, heads, cs, sgn);
  else hipLaunchKernelGGL((k_rope_apply<float>), dim3(blocks), dim3(256), 0, s, (float*)x, ld, rows, heads, cs, sgn);
  TTV_CHECK_LAUNCH("rope_apply");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Codebook usage histogram (codebook_logging.py:19-25: sum of bincounts).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_histogram(const int* __restrict__ idx, int n, unsigned long long* __restrict__ counts,
                                                   int size) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int v = idx[i];
    if (v >= 0 && v < size) atomicAdd(&counts[v], 1ULL);
  }
}

int ttvk_histogram(const int* idx, int n, int64_t* counts, int size, hipStream_t s) {
  if (n == 0) return TTV_OK;
  int blocks = ttv_cdiv(n, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_histogram, dim3(blocks), dim3(256), 0, s, idx, n, (unsigned long long*)counts, size);
  TTV_CHECK_LAUNCH("histogram");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of the constant rows v[f] = m * rsqrt(m^2+eps) * gain[f] (k_fill_const_rows), given colsum[f] = sum over those
// rows of the incoming gradient:  dgain[f] += colsum[f] * m*rstd ;  dmask += sum_f colsum[f]*gain[f] * eps*(m^2+eps)^-1.5
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_const_rows_bwd(const float* __restrict__ colsum, const float* __restrict__ mask_token,
                                                        const float* __restrict__ gain, float eps, float* __restrict__ dgain,
                                                        float* __restrict__ dmask, int d) {
  __shared__ float red[4];
  const float m = round_to<T>(mask_token[0]);
  const float rstd = 1.0f / sqrtf(m * m + eps);
  float acc = 0.f;
  for (int f = threadIdx.x; f < d; f += 256) {
    atomicAdd(dgain + f, colsum[f] * m * rstd);
    acc += colsum[f] * gain[f];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dmask, (red[0] + red[1] + red[2] + red[3]) * eps * rstd * rstd * rstd);
}

int ttvk_const_rows_bwd(const float* colsum, const float* mask_token, const float* gain, int dt, float eps, float* dgain, float* dmask,
                        int d, hipStream_t s) {
  if (!dgain && !dmask) return TTV_OK;
  if (dt == TTV_BF16) hipLaunchKernelGGL((k_const_rows_bwd<bf16_t>), dim3(1), dim3(256), 0, s, colsum, mask_token, gain, eps, dgain, dmask, d);
  else hipLaunchKernelGGL((k_const_rows_bwd<float>), dim3(1), dim3(256), 0, s, colsum, mask_token, gain, eps, dgain, dmask, d);
  TTV_CHECK_LAUNCH("const_rows_bwd");
  return TTV_OK;
}

// ------------------------------------------------------------------------------------------------
// RoPE table of a packed batch, gathered on the device from the host-evaluated base table
// base_cos/base_sin fp32 [n_ids, F] = cos/sin(inv_freq[f] * n) (fp64 on the host, rope.py:40-54).  Row ids (rope.py:59-67):
// latent i -> (i,i,i); patch (t,h,w) -> (t,h,w) + K.  out [L,64] = cos[32] | sin[32], column f*3 + axis, tail (1, 0).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rope_build(const float* __restrict__ base_cos, const float* __restrict__ base_sin, int n_ids,
                                                    int F, const int* __restrict__ clip_desc, const int* __restrict__ cu,
                                                    const int* __restrict__ row_seq, float* __restrict__ out, int total_rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total_rows * 32) return;
  const int row = i >> 5, j = i & 31;
  const int b = row_seq[row];
  const int* ds = clip_desc + b * 8;
  const int gh = ds[4], gw = ds[5];
  const int P = ds[3] * gh * gw, S = cu[b + 1] - cu[b], K = S - P;
  const int local = row - cu[b];
  float c = 1.f, sn = 0.f;
  if (j < 3 * F) {
    const int f = j / 3, axis = j - 3 * f;
    int id;
    if (local < K) id = local;
    else {
      const int p = local - K;
      const int w = p % gw, h = (p / gw) % gh, t = p / (gw * gh);
      id = (axis == 0 ? t : axis == 1 ? h : w) + K;
    }
    id = id < n_ids ? id : n_ids - 1;
    c = base_cos[id * F + f];
    sn = base_sin[id * F + f];
  }
  out[(size_t)row * 64 + j] = c;
  out[(size_t)row * 64 + 32 + j] = sn;
}

int ttvk_rope_build(const float* base_cos, const float* base_sin, int n_ids, int F, const int* clip_desc, const int* cu, const int* row_seq,
                    float* out, int total_rows, hipStream_t s) {
  if (total_rows == 0) return TTV_OK;
  hipLaunchKernelGGL(k_rope_build, dim3(ttv_cdiv(total_rows * 32, 256)), dim3(256), 0, s, base_cos, base_sin, n_ids, F, clip_desc, cu, row_seq, out, total_rows);
  TTV_CHECK_LAUNCH("rope_build");
  return TTV_OK;
}


// ------------------------------------------------------------------------------------------------
// Loader tail on the device (reference dataset/video_dataset.py:116-119: v2.ToDtype(scale=True) + Normalize(0.5, 0.5) after the
// frames were permuted to channel-first): decoded frames uint8 [T,H,W,3] -> clip [3,T,H,W] in dtype, value u8 / 127.5 - 1 in
// fp32 (one correctly rounded division, one subtraction: bit-equal to the torch expression the host loader used), rounded once to
// dtype.  HBM-bound: 3 bytes in, 3 elements out per pixel; a thread takes 4 consecutive pixels (12 bytes in, 3 x 4 elements out).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_clip_from_u8(const uint8_t* __restrict__ frames, T* __restrict__ clip, long long n_pix) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;      // group of 4 pixels
  const long long p0 = q * 4;
  if (p0 >= n_pix) return;
  if (p0 + 4 <= n_pix) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(frames + p0 * 3);     // 12 bytes, 4-byte aligned (p0 % 4 == 0)
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    const uint8_t b[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24), (uint8_t)w1, (uint8_t)(w1 >> 8),
                           (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24), (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __fsub_rn(__fdiv_rn((float)b[3 * i + c], 127.5f), 1.0f);
      Vec4<T>::store(clip + (size_t)c * n_pix + p0, v);
    }
  } else {
    for (long long p = p0; p < n_pix; ++p)
      for (int c = 0; c < 3; ++c) clip[(size_t)c * n_pix + p] = Cvt<T>::from_f(__fsub_rn(__fdiv_rn((float)frames[p * 3 + c], 127.5f), 1.0f));
  }
}

int ttvk_clip_from_u8(const void* frames, long long n_pix, void* clip, int dtype, hipStream_t s) {
  if (n_pix == 0) return TTV_OK;
  TTV_CHECK_ARG((uintptr_t)frames % 4 == 0 && (uintptr_t)clip % 16 == 0 && n_pix % 4 == 0, "clip_from_u8: frames 4-byte, clip 16-byte aligned, T*H*W %% 4 == 0");
  const long long groups = (n_pix + 3) / 4;
  dim3 grid((unsigned)((groups + 255) / 256));
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_clip_from_u8<bf16_t>), grid, dim3(256), 0, s, (const uint8_t*)frames, (bf16_t*)clip, n_pix);
  else if (dtype == TTV_F32) hipLaunchKernelGGL((k_clip_from_u8<float>), grid, dim3(256), 0, s, (const uint8_t*)frames, (float*)clip, n_pix);
  else { ttv_set_error("clip_from_u8: bad dtype"); return TTV_ERR_INVALID; }
  TTV_CHECK_LAUNCH("clip_from_u8");
  return TTV_OK;
}

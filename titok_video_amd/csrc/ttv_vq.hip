// Nearest-codebook-entry (L2) vector quantiser: indices[r] = argmin_n || z[r] - codebook[n] ||^2, lowest index on ties.
//
// Not a reference component: the reference's only quantiser is FSQ (model/quantizer/fsq.py:78-135).  BASELINE.json's north_star
// asks for the L2 formulation as well ("nearest-codebook-entry L2 distance + straight-through lookup ... LDS-staged codebook
// chunks ... wavefront shuffle/butterfly reductions for argmin") and its configs #4/#5 name synthetic 8192x32 / 16384x64
// codebooks.  It is pinned two ways (tests/test_hip_vq.py): on the FSQ lattice  implicit_codebook * (levels // 2)  (fsq.py:73-76)
// applied to the bounded vector it returns FSQ's own indices away from rounding ties, and on random codebooks it equals a float64
// cdist + argmin oracle (oracle/vq_oracle.py) away from distance ties.
//
// argmin_n ||z - c_n||^2 = argmax_n ( z . c_n - ||c_n||^2 / 2 ): a GEMM with a bias, so the distances ride on the matrix cores.
//   block = 4 waves x 32 rows of z; the codebook streams through LDS in chunks of 128 entries (register-staged, the next chunk's
//   loads in flight behind the MFMAs of the current one, rows padded by one element: conflict-free column reads);
//   per 32-entry sub-tile:  S^T[entry][row] = C Z^T  with the ENTRY on the MFMA row and the z ROW on the lane, accumulator
//   preloaded with -||c||^2/2  ->  a lane holds 16 scores of ONE row, so the running (best score, best index) is lane-local;
//   float32 input: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32; products and sums in fp32), bf16 input: v_mfma_f32_32x32x16_bf16.
//   A sub-tile's maximum is found with 8 v_max3_f32; the 16-way index search runs only when some lane's maximum beats its running
//   best (after the first chunks that is rare), so the steady state costs ~10 vector instructions per 32 x 32 scores.
//   The two lanes that share a row (lane, lane ^ 32: entries 4h + .. of every 8) merge with one xor-32 exchange - the butterfly
//   has a single level in this layout - preferring the lower index on equal scores.
//   Grid = row tiles x codebook SPLITS (a 128-row tile per block alone is one block per CU at 32 k rows): every split scans its
//   share of the entries and merges into a per-row 64-bit key (ordered score << 32 | ~index) with one atomicMax per row, which
//   keeps the lowest index among equal scores; k_vq_finish unpacks the keys.
#include "ttv_common.h"
#include "ttv_kernels.h"

#define VQ_CH 128   // codebook entries per LDS chunk

// ||c_n||^2 per entry, fp32 (one wave per 64 entries; C is small)
template <typename T>
__global__ __launch_bounds__(256) void k_vq_norms(const T* __restrict__ cb, int ld, int N, int C, float* __restrict__ cnorm) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = Cvt<T>::to_f(cb[(size_t)n * ld + c]);
    s = fmaf(v, v, s);
  }
  cnorm[n] = s;
}

// KS = MFMA k-steps per sub-tile: fp32: ceil(C / 2) of depth 2; bf16: ceil(C / 16) of depth 16.  CP = padded LDS row length (elements).
template <typename T, int CMAX>
__global__ __launch_bounds__(256, 2) void k_vq_l2_argmin(const T* __restrict__ z, int ldz, const T* __restrict__ cb, int ldc,
                                                         const float* __restrict__ cnorm, int rows, int N, int C,
                                                         unsigned long long* __restrict__ keys, int n_split) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int KD = F32 ? 2 : 16;                 // k depth of one MFMA
  constexpr int KS = (CMAX + KD - 1) / KD;         // k-steps
  constexpr int CW = KS * KD;                      // dims per row incl. zero padding
  constexpr int CP = F32 ? CW + 1 : CW + 8;        // LDS row stride: fp32 odd (column reads by 32 lanes hit 32 banks); bf16 +16 B
  __shared__ __attribute__((aligned(16))) T cl[2][VQ_CH * CP];
  __shared__ float nl[2][VQ_CH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rtile = blockIdx.x / n_split, split = blockIdx.x - rtile * n_split;
  const int row = rtile * 128 + wave * 32 + r;
  const int rowc = row < rows ? row : rows - 1;
  // this block's share of the codebook: chunks [ch0, ch1)
  const int nch_all = (N + VQ_CH - 1) / VQ_CH;
  const int ch0 = (int)((long)nch_all * split / n_split), ch1 = (int)((long)nch_all * (split + 1) / n_split);
  if (ch0 >= ch1) return;

  // z fragments (B operand): fp32: zf[s] = z[row][2s + h]; bf16: zb[s] = z[row][16s + 8h .. +7]
  float zf[F32 ? KS : 1];
  bf16x8 zb[F32 ? 1 : KS];
  if constexpr (F32) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 2 * s + h;
      zf[s] = c < C ? (float)z[(size_t)rowc * ldz + c] : 0.f;
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 16 * s + 8 * h + j;
        zb[s][j] = c < C ? z[(size_t)rowc * ldz + c] : (bf16_t)0.f;
      }
  }

  // chunk staging in 16-byte vectors (VEC elements): vector v of entry e for e * (CW / VEC) + v = tid + 256 i; dims beyond C and
  // entries beyond N are zero (a padded entry gets norm +inf and can never win).  Rows whose C or leading dimension is not a
  // multiple of VEC are gathered element by element into the same vectors.
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int VPR = CW / VEC;                    // vectors per padded row (CW is a multiple of VEC for both dtypes... see below)
  static_assert(CW % VEC == 0, "padded row must be whole vectors");
  constexpr int PER_T = (VQ_CH * VPR + 255) / 256;
  uint4 st[PER_T];
  float sn = 0.f;
  const bool vec_ok = (C % VEC == 0) && (ldc % VEC == 0) && ((uintptr_t)cb % 16 == 0);
#define VQ_GLOAD(n0_)                                                                             \
  do {                                                                                            \
    _Pragma("unroll") for (int i__ = 0; i__ < PER_T; ++i__) {                                     \
      const int f__ = tid + 256 * i__, e__ = f__ / VPR, c__ = (f__ - e__ * VPR) * VEC;            \
      const int n__ = (n0_) + e__;                                                                \
      uint4 v__ = {0u, 0u, 0u, 0u};                                                               \
      if (e__ < VQ_CH && n__ < N) {                                                               \
        if (vec_ok) {                                                                             \
          if (c__ < C) v__ = *reinterpret_cast<const uint4*>(cb + (size_t)n__ * ldc + c__);       \
        } else {                                                                                  \
          T tmp__[VEC];                                                                           \
          _Pragma("unroll") for (int j__ = 0; j__ < VEC; ++j__)                                   \
            tmp__[j__] = (c__ + j__) < C ? cb[(size_t)n__ * ldc + c__ + j__] : (T)0.f;            \
          v__ = *reinterpret_cast<const uint4*>(tmp__);                                           \
        }                                                                                         \
      }                                                                                           \
      st[i__] = v__;                                                                              \
    }                                                                                             \
    if (tid < VQ_CH) sn = ((n0_) + tid) < N ? -0.5f * cnorm[(n0_) + tid] : -INFINITY;             \
  } while (0)
#define VQ_LSTORE(buf_)                                                                           \
  do {                                                                                            \
    _Pragma("unroll") for (int i__ = 0; i__ < PER_T; ++i__) {                                     \
      const int f__ = tid + 256 * i__, e__ = f__ / VPR, c__ = (f__ - e__ * VPR) * VEC;            \
      if (e__ < VQ_CH) {                                                                          \
        if (F32) { /* odd row stride: rows are not 16-byte aligned, store the four elements */    \
          const float* p__ = reinterpret_cast<const float*>(&st[i__]);                            \
          _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) reinterpret_cast<float*>(&cl[buf_][0])[e__ * CP + c__ + j__] = p__[j__]; \
        } else {                                                                                  \
          *reinterpret_cast<uint4*>(&cl[buf_][e__ * CP + c__]) = st[i__];                         \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
    if (tid < VQ_CH) nl[buf_][tid] = sn;                                                          \
  } while (0)

  float best = -INFINITY;
  int bidx = 0;
  VQ_GLOAD(ch0 * VQ_CH);
  VQ_LSTORE(0);
  __syncthreads();
  for (int ch = ch0; ch < ch1; ++ch) {
    const int buf = (ch - ch0) & 1;
    if (ch + 1 < ch1) VQ_GLOAD((ch + 1) * VQ_CH);
#pragma unroll
    for (int t = 0; t < VQ_CH / 32; ++t) {
      // accumulator preloaded with -||c||^2 / 2 of the lane's 16 entries: entry (e & 3) + 8 (e >> 2) + 4 h of the sub-tile
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {       // four consecutive entries per 16-byte read (same address in every lane of a half: broadcast)
        const f32x4 nb = *reinterpret_cast<const f32x4*>(&nl[buf][t * 32 + 8 * g + 4 * h]);
        acc[4 * g] = nb[0]; acc[4 * g + 1] = nb[1]; acc[4 * g + 2] = nb[2]; acc[4 * g + 3] = nb[3];
      }
      const T* crow = &cl[buf][(t * 32 + r) * CP];
      if constexpr (F32) {
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)crow[2 * s + h], zf[s], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(crow + 16 * s + 8 * h);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zb[s], acc, 0, 0, 0);
        }
      }
      // sub-tile maximum; the index search only where it beats the running best (wave-uniform branch)
      float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
      for (int e = 3; e < 15; e += 2) m = fmaxf(fmaxf(m, acc[e]), acc[e + 1]);
      m = fmaxf(m, acc[15]);
      if (__builtin_amdgcn_ballot_w64(m > best) != 0ull) {
        const int n0 = ch * VQ_CH + t * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {       // increasing entry index within the lane: strict > keeps the earlier entry on ties
          const bool up = acc[e] > best;
          best = up ? acc[e] : best;
          bidx = up ? n0 + (e & 3) + 8 * (e >> 2) : bidx;
        }
      }
    }
    if (ch + 1 < ch1) VQ_LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef VQ_GLOAD
#undef VQ_LSTORE
  // merge the two lanes of a row (xor-32 butterfly level): larger score, lower index on equal scores
  const float ob = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(bidx, 32, 64);
  if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  if (row < rows && h == 0) {
    // order-preserving map of the fp32 score to uint32 (negative values flipped), then ~index so that among equal scores the
    // LOWER index is the larger key
    const unsigned u = __builtin_bit_cast(unsigned, best);
    const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    atomicMax(&keys[row], ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bidx));
  }
}

// keys -> indices (+ squared distances ||z||^2 - 2 score)
template <typename T>
__global__ __launch_bounds__(256) void k_vq_finish(const unsigned long long* __restrict__ keys, const T* __restrict__ z, int ldz, int rows, int C,
                                                   int* __restrict__ indices, float* __restrict__ best_dist) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const unsigned long long k = keys[row];
  indices[row] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
  if (best_dist) {
    const unsigned ord = (unsigned)(k >> 32);
    const unsigned u = (ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord;
    const float score = __builtin_bit_cast(float, u);
    float zz = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = Cvt<T>::to_f(z[(size_t)row * ldz + c]);
      zz = fmaf(v, v, zz);
    }
    best_dist[row] = zz - 2.0f * score;
  }
}

// straight-through lookup: codes[r] = codebook[indices[r]]
template <typename T>
__global__ __launch_bounds__(256) void k_vq_lookup(const T* __restrict__ cb, int ldc, const int* __restrict__ indices, int rows, int C,
                                                   T* __restrict__ codes, int ldo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * C) return;
  const int rr = (int)(i / C), c = (int)(i - (long)rr * C);
  codes[(size_t)rr * ldo + c] = cb[(size_t)indices[rr] * ldc + c];
}

int ttvk_vq_norms(const void* cb, int dtype, int ld, int N, int C, float* cnorm, hipStream_t s) {
  if (N == 0) return TTV_OK;
  TTV_CHECK_ARG(dtype == TTV_BF16 || dtype == TTV_F32, "vq_norms: bad dtype");
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_vq_norms<bf16_t>), dim3(ttv_cdiv(N, 256)), dim3(256), 0, s, (const bf16_t*)cb, ld, N, C, cnorm);
  else hipLaunchKernelGGL((k_vq_norms<float>), dim3(ttv_cdiv(N, 256)), dim3(256), 0, s, (const float*)cb, ld, N, C, cnorm);
  TTV_CHECK_LAUNCH("vq_norms");
  return TTV_OK;
}

int64_t ttvk_vq_workspace_bytes(int rows) { return (int64_t)rows * 8; }

int ttvk_vq_l2_argmin(const void* z, int dtype, int ldz, const void* cb, int ldc, const float* cnorm, int rows, int N, int C, int* indices,
                      float* best_dist, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(dtype == TTV_BF16 || dtype == TTV_F32, "vq_l2_argmin: bad dtype");
  TTV_CHECK_ARG(N >= 1 && C >= 1 && C <= 64, "vq_l2_argmin: codebook dim %d unsupported (1..64)", C);
  TTV_CHECK_ARG(workspace && workspace_bytes >= ttvk_vq_workspace_bytes(rows) && (uintptr_t)workspace % 8 == 0, "vq_l2_argmin: workspace too small");
  unsigned long long* keys = (unsigned long long*)workspace;
  if (hipMemsetAsync(keys, 0, (size_t)rows * 8, s) != hipSuccess) { ttv_set_error("vq_l2_argmin: memset failed"); return TTV_ERR_LAUNCH; }
  // enough blocks for ~4 per CU: split the codebook when there are few row tiles (never finer than one 128-entry chunk)
  const int rt = ttv_cdiv(rows, 128), nch = ttv_cdiv(N, VQ_CH);
  int n_split = ttv_cdiv(1024, rt);
  n_split = n_split < 1 ? 1 : (n_split > nch ? nch : n_split);
  dim3 grid(rt * n_split);
#define VQ_LAUNCH(T_, CM_) hipLaunchKernelGGL((k_vq_l2_argmin<T_, CM_>), grid, dim3(256), 0, s, (const T_*)z, ldz, (const T_*)cb, ldc, cnorm, rows, N, C, keys, n_split)
  if (dtype == TTV_F32) {
    if (C <= 8) VQ_LAUNCH(float, 8); else if (C <= 32) VQ_LAUNCH(float, 32); else VQ_LAUNCH(float, 64);
  } else {
    if (C <= 16) VQ_LAUNCH(bf16_t, 16); else if (C <= 32) VQ_LAUNCH(bf16_t, 32); else VQ_LAUNCH(bf16_t, 64);
  }
#undef VQ_LAUNCH
  TTV_CHECK_LAUNCH("vq_l2_argmin");
  if (dtype == TTV_F32) hipLaunchKernelGGL((k_vq_finish<float>), dim3(ttv_cdiv(rows, 256)), dim3(256), 0, s, keys, (const float*)z, ldz, rows, C, indices, best_dist);
  else hipLaunchKernelGGL((k_vq_finish<bf16_t>), dim3(ttv_cdiv(rows, 256)), dim3(256), 0, s, keys, (const bf16_t*)z, ldz, rows, C, indices, best_dist);
  TTV_CHECK_LAUNCH("vq_finish");
  return TTV_OK;
}

int ttvk_vq_lookup(const void* cb, int dtype, int ldc, const int* indices, int rows, int C, void* codes, int ldo, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(dtype == TTV_BF16 || dtype == TTV_F32, "vq_lookup: bad dtype");
  const long n = (long)rows * C;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TTV_BF16) hipLaunchKernelGGL((k_vq_lookup<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)cb, ldc, indices, rows, C, (bf16_t*)codes, ldo);
  else hipLaunchKernelGGL((k_vq_lookup<float>), grid, dim3(256), 0, s, (const float*)cb, ldc, indices, rows, C, (float*)codes, ldo);
  TTV_CHECK_LAUNCH("vq_lookup");
  return TTV_OK;
}

// d codebook[idx[r], :] += d codes[r, :]: the gradient of the straight-through lookup codes = codebook[idx] with respect to the
// codebook (fp32 accumulation; entries shared by several rows add up: float atomics, a few KB of traffic).
template <typename T>
__global__ __launch_bounds__(256) void k_vq_lookup_bwd(const T* __restrict__ dcodes, int ld, const int* __restrict__ indices, int rows, int C,
                                                       float* __restrict__ dcb, int ldc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * C) return;
  const int r = (int)(i / C), c = (int)(i % C);
  atomicAdd(dcb + (size_t)indices[r] * ldc + c, Cvt<T>::to_f(dcodes[(size_t)r * ld + c]));
}

int ttvk_vq_lookup_bwd(const void* dcodes, int dtype, int ld, const int* indices, int rows, int C, float* dcb, int ldc, hipStream_t s) {
  if (rows == 0) return TTV_OK;
  TTV_CHECK_ARG(dtype == TTV_BF16 || dtype == TTV_F32, "vq_lookup_backward: bad dtype");
  dim3 grid((unsigned)(((long)rows * C + 255) / 256));
  if (dtype == TTV_F32) hipLaunchKernelGGL((k_vq_lookup_bwd<float>), grid, dim3(256), 0, s, (const float*)dcodes, ld, indices, rows, C, dcb, ldc);
  else hipLaunchKernelGGL((k_vq_lookup_bwd<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dcodes, ld, indices, rows, C, dcb, ldc);
  TTV_CHECK_LAUNCH("vq_lookup_bwd");
  return TTV_OK;
}

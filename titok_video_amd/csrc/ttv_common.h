// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/titok_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TTV_WAVE 64

// ---- error plumbing (host) ------------------------------------------------------------------
void ttv_set_error(const char* fmt, ...);
#define TTV_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      ttv_set_error(__VA_ARGS__);         \
      return TTV_ERR_INVALID;             \
    }                                     \
  } while (0)
#define TTV_CHECK_LAUNCH(what)                                            \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      ttv_set_error("%s: %s", what, hipGetErrorString(e__));              \
      return TTV_ERR_LAUNCH;                                              \
    }                                                                     \
  } while (0)

// ---- storage <-> fp32 -----------------------------------------------------------------------
template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe
};
// round a fp32 value through the storage type (what a store+load of T would do)
template <typename T> __device__ __forceinline__ float round_to(float v) { return Cvt<T>::to_f(Cvt<T>::from_f(v)); }

// 4 consecutive elements
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
    bf16x4 b = *reinterpret_cast<const bf16x4*>(p);
    f32x4 v = {(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    return v;
  }
  static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
    bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = b;
  }
};

// ---- wave reductions (64 lanes, xor butterfly 32, 16, 8, 4, 2, 1) -----------------------------------
// Every exchange on the vector ALU: v_permlane32_swap / v_permlane16_swap for the two cross-row steps, DPP moves inside the 16-lane
// rows (row_ror:8 = xor 8; xor 4 as two bank-masked rotations; quad_perm for xor 2 and xor 1).  __shfl_xor lowers to ds_bpermute_b32 +
// s_waitcnt lgkmcnt(0) - an LDS round trip per step, six in a row per reduction, and the row kernels (RMSNorm forward / backward, row
// statistics, fp8 row quantisation) are chains of such reductions.  Same partners in the same order as the butterfly above: the sums
// are the bits they were (operands of an addition may swap sides, which changes nothing).
__device__ __forceinline__ float wave_xor_dpp8(float v) {      // value of lane ^ 8
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x128, 0xF, 0xF, true));       // row_ror:8
}
__device__ __forceinline__ float wave_xor_dpp4(float v) {      // value of lane ^ 4: banks 0, 2 take lane + 4 (row_ror:12), banks 1, 3 lane - 4 (row_ror:4)
  const int i = __builtin_bit_cast(int, v);
  int r = __builtin_amdgcn_update_dpp(0, i, 0x12C, 0xF, 0x5, false);      // row_ror:12 -> lane i reads lane (i - 12) & 15 = (i + 4) & 15
  r = __builtin_amdgcn_update_dpp(r, i, 0x124, 0xF, 0xA, false);          // row_ror:4  -> lane i reads lane (i - 4) & 15
  return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float wave_xor_dpp2(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float wave_xor_dpp1(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float wave_sum(float v) {
  {
    const unsigned u = __float_as_uint(v);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned b0 = b[0], b1 = b[1];
    v = __uint_as_float(b0) + __uint_as_float(b1);
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    v = __uint_as_float(a0) + __uint_as_float(a1);
  }
  v += wave_xor_dpp8(v);
  v += wave_xor_dpp4(v);
  v += wave_xor_dpp2(v);
  v += wave_xor_dpp1(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  {
    const unsigned u = __float_as_uint(v);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned b0 = b[0], b1 = b[1];
    v = fmaxf(__uint_as_float(b0), __uint_as_float(b1));
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    v = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
  }
  v = fmaxf(v, wave_xor_dpp8(v));
  v = fmaxf(v, wave_xor_dpp4(v));
  v = fmaxf(v, wave_xor_dpp2(v));
  v = fmaxf(v, wave_xor_dpp1(v));
  return v;
}

// ---- lane exchanges on the vector ALU (gfx950 v_permlane16_swap / v_permlane32_swap) --------------------------------------------
// hipcc lowers __shfl_xor(v, 16 | 32, 64) to index arithmetic + ds_bpermute_b32 + s_waitcnt lgkmcnt(0): an LDS round trip per
// exchange, serialised by its wait.  The swaps below are single VALU instructions.  (Results are read through named unsigneds:
// __builtin_bit_cast applied to a subscript of the builtin's vector result reads element 0 for both - hipcc 7.2.)
// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (same association as ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32))
__device__ __forceinline__ float quad16_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // rows (r0, r0, r2, r2) | (r1, r1, r3, r3)
  const unsigned a0 = a[0], a1 = a[1];
  const float s = __uint_as_float(a0) + __uint_as_float(a1);
  const unsigned t = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(t, t, false, false);      // halves (lo, lo) | (hi, hi)
  const unsigned b0 = b[0], b1 = b[1];
  return __uint_as_float(b0) + __uint_as_float(b1);
}
__device__ __forceinline__ float quad16_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const float s = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
  const unsigned t = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(t, t, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return fmaxf(__uint_as_float(b0), __uint_as_float(b1));
}
// sum over the two lanes l, l ^ 32
__device__ __forceinline__ float pair32_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return __uint_as_float(b0) + __uint_as_float(b1);
}
// Lanes l and l ^ 16 (16-lane rows 2k and 2k + 1) each hold two 8-byte groups (p0, p1) of a row of which the even lane stores
// (own p0, partner's p0) and the odd lane (partner's p1, own p1) as one 16-byte vector: v_permlane16_swap(vdst = p0, src = p1)
// leaves exactly that pair in (vdst, src) for both.
__device__ __forceinline__ uint4 xchg16_pair(uint2 p0, uint2 p1) {
  const auto x = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
  const auto y = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
  const unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
  return make_uint4(x0, y0, x1, y1);
}

// ---- optional HIP-event bracketing of one kernel class (ttv_prof_begin / ttv_prof_end) ----------------
extern int g_ttv_prof_class;
extern thread_local int g_ttv_debug;
extern long long* g_ttv_stamps;
struct TtvProfScope {
  int slot;
  hipStream_t s;
  TtvProfScope(int cls, hipStream_t stream);
  ~TtvProfScope();
};

static inline int ttv_cdiv(int a, int b) { return (a + b - 1) / b; }
// bf16 path: gelu(g) * x without erff().  Phi(g) = sigmoid(p(g)), p odd of degree 5 (minimax fit, g clamped to +-8 so the
// negative g^5 coefficient never takes over): max |g*Phi - gelu_erf(g)| = 2.6e-5 over all g - 1/10 of a bf16 half-ulp at
// 0.06 - and 12 VALU issue slots per element (v_exp_f32 / v_rcp_f32 count 2 each) against 18 for an erf polynomial of
// the same accuracy; single-lane ops on purpose: v_pk_*_f32 beside MFMAs costs more than the two scalar ops it replaces
// (measured: the packed form of this epilogue ran 1.4x longer).  The fp32 parity path uses erff().
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float geglu_fast(float g, float x) {
  const float gc = __builtin_amdgcn_fmed3f(g, -8.0f, 8.0f);
  const float g2 = gc * gc;
  float q = fmaf(g2, 1.014262858e-03f, -1.067757308e-01f);       // -log2(e) * (c2 g^2 + c1)
  q = fmaf(q, g2, -2.301121361e+00f);                            // -log2(e) * c0
  const float e = __builtin_amdgcn_exp2f(q * gc);                // exp(-p(g))
  return (g * __builtin_amdgcn_rcpf(1.0f + e)) * x;
}
__device__ __forceinline__ f32x2 geglu_pair_fast(f32x2 g, f32x2 x) { return (f32x2){geglu_fast(g.x, x.x), geglu_fast(g.y, x.y)}; }

// Eight independent geglu_fast() evaluations written stage by stage (clamp/polynomial, exp, 1+e, rcp, products) with
// scheduling fences in between: the compiler otherwise interleaves only two dependency chains, and a lone wave then
// stalls on the v_exp_f32 / v_rcp_f32 result latency at every step of every chain.
__device__ __forceinline__ void geglu_fast8(const float (&g)[8], const float (&x)[8], float (&h)[8]) {
  float a[8], e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float gc = __builtin_amdgcn_fmed3f(g[k], -8.0f, 8.0f);
    const float g2 = gc * gc;
    float q = fmaf(g2, 1.014262858e-03f, -1.067757308e-01f);
    q = fmaf(q, g2, -2.301121361e+00f);
    a[k] = q * gc;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = __builtin_amdgcn_exp2f(a[k]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = 1.0f + e[k];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = __builtin_amdgcn_rcpf(e[k]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) h[k] = (g[k] * e[k]) * x[k];
}

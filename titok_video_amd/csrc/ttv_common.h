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

// ---- wave reductions (64 lanes, xor butterfly) ----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- optional HIP-event bracketing of one kernel class (ttv_prof_begin / ttv_prof_end) ----------------
extern int g_ttv_prof_class;
extern int g_ttv_debug;
struct TtvProfScope {
  int slot;
  hipStream_t s;
  TtvProfScope(int cls, hipStream_t stream);
  ~TtvProfScope();
};

static inline int ttv_cdiv(int a, int b) { return (a + b - 1) / b; }
// bf16 path: gelu(g) * x for two elements at once, transcendental-free so that it runs on the packed-fp32 VALU
// (v_pk_fma_f32: 2 lanes-worth per issue).  Phi(v) - 1/2 = t * Q(u), t = clamp(v, -5, 5) / 5, u = 2 t^2 - 1 in [-1, 1],
// Q a degree-10 minimax fit (coefficients O(1): no cancellation in fp32 Horner).  max |gelu - exact| = 1.1e-5 over all v,
// i.e. 1/20 of a bf16 half-ulp at 0.03; the fp32 parity path uses erff().
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 geglu_pair_fast(f32x2 g, f32x2 x) {
  const f32x2 vc = {__builtin_amdgcn_fmed3f(g.x, -5.0f, 5.0f), __builtin_amdgcn_fmed3f(g.y, -5.0f, 5.0f)};
  const f32x2 t = vc * 0.2f;
  const f32x2 u = __builtin_elementwise_fma(t + t, t, (f32x2)(-1.0f));
  f32x2 q = (f32x2)(1.138652562e-02f);
  q = __builtin_elementwise_fma(q, u, (f32x2)(-3.139007089e-02f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(3.510615383e-02f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(-4.164913603e-02f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(7.773959393e-02f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(-1.210758038e-01f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(1.587207418e-01f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(-2.015958492e-01f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(2.574400549e-01f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(-3.515025932e-01f));
  q = __builtin_elementwise_fma(q, u, (f32x2)(7.068215094e-01f));
  const f32x2 phi = __builtin_elementwise_fma(t, q, (f32x2)(0.5f));
  return (g * phi) * x;
}



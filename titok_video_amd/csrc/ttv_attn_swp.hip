// Software-pipelined bf16 attention for tables of FULL items with pre-scaled q (inference; replaces flash_attn_varlen_func at
// reference model/base/transformer.py:100 and the sigmoid gate at :103 like k_attn_bf16, whose decomposition it keeps:
// block = 4 waves = 128 query rows of one (sequence, q-head), 32 queries per wave, 64-key K / V tiles by LDS-DMA).
//
// Why another kernel (round 5).  On one SIMD the matrix pipe and the vector issue port overlap only for instructions of the SAME
// wave that sit behind an MFMA in program order (profiles/r04_pingpong.txt, r04_mfma_shapes.txt): k_attn_bf16's phases
//   S MFMAs -> exp2 / pack / row sums -> PV MFMAs
// therefore cost a SIMD the SUM of their issue times (885 cycles per 32-query x 64-key unit for 512 cycles of MFMAs), however many
// waves it holds.  Here every MFMA of the key loop carries the vector work of a NEIGHBOURING tile behind it:
//
//   phase X(t):  8 MFMAs  S(t+1) = K(t+1) Q^T   |  exp2 / pack / row sums of keys 32-63 of tile t   |  16 V(t) fragment reads
//   check(t):    row sums of tile t in range?  (else: rare path, below)
//   phase Y(t):  8 MFMAs  O += V(t)^T P(t)^T    |  exp2 / pack / row sums of keys  0-31 of tile t+1 |   8 K(t+2) fragment reads
//
// per MFMA exactly 2 v_exp_f32 + 1 v_cvt_pk_bf16_f32 + 2 v_add_f32 (issue cost 8 + 2 x 8.7 + 3 x 3 = 34 cycles against the MFMA's
// 32: tools/ubench/valu_rates.hip), pinned per gap with sched_group_barrier.  What makes the mix that thin:
//   * q arrives multiplied by scale * log2(e) (ttv_layer_weights.qkv_q_prescaled) and the score accumulators START from -m (the
//     running reference of the row, replicated over a 16-register vector that is the C operand of each chain's first MFMA): a
//     score leaves the matrix pipe as the exponent, no multiply, no subtract;
//   * no row maximum per tile: the reference m is the maximum of the row's first 64 scores and stays unless a tile's row sum
//     leaves [0, 2^30] (or is NaN) - any reference gives the same quotient, bf16 P keeps 8 significant bits at any magnitude
//     and O / l are fp32.  The raw scores of tile t stay in registers until check(t) (exp2 writes temporaries), so the rare path
//     takes their true maximum, shifts O, l, S(t), S(t+1) and the start vector, and redoes the tile's P: nothing is recomputed
//     from memory;
//   * row sums are plain v_add_f32 chains (v_pk_add_f32 beside MFMAs costs more than the two adds it replaces).
// Registers: S(t) 32 + S(t+1) 32 + P 16..24 + O 32 + K fragments 32 + V fragments 32 + q 16 + start vector 16 + addresses: two
// waves per SIMD (launch bounds 256 x 2), 32 KB LDS per block (two 2-slot rings).
//
// LDS hand-over: ONE barrier per tile at the top of iteration t.  Behind it every wave has finished X(t-1) / Y(t-1), i.e. its
// reads of V(t-1) and K(t+1): V(t+1) is issued into V(t-1)'s slot and K(t+3) into K(t+1)'s; both are waited for (vmcnt(0): a
// full iteration later, nothing to wait for in practice) in front of the next barrier, where V(t+1) is first read (X(t+1)) and
// K(t+3) an iteration and a half later (Y(t+1)).
#include <stdlib.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

#define SWP_KB 64
#define SWP_BIG 1073741824.0f      // 2^30: a tile's row sum beyond it (or NaN) sends the wave through the rare path

typedef __attribute__((address_space(3))) bf16x4 swp_lds_bf16x4;
__device__ __forceinline__ bf16x4 swp_read_tr16(const char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((swp_lds_bf16x4*)(lds_ptr));
}

#ifdef SWP_STAMPS
#define SWP_STAMP_DECL unsigned long long st_prev__ = 0, st_acc__[6] = {0, 0, 0, 0, 0, 0}
#define SWP_STAMP_START()                                                                              \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev__)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
#define SWP_STAMP(seg_)                                                                                \
  do {                                                                                                 \
    unsigned long long t__;                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    st_acc__[seg_] += t__ - st_prev__;                                                                 \
    st_prev__ = t__;                                                                                   \
  } while (0)
#else
#define SWP_STAMP_DECL
#define SWP_STAMP_START()
#define SWP_STAMP(seg_)
#endif

template <bool GATE>
__global__ __launch_bounds__(256, 2) void k_attn_swp(const bf16_t* __restrict__ qkvg, int ld, bf16_t* __restrict__ out, int ldo,
                                                     const int* __restrict__ cu, const int* __restrict__ qblocks, int d_model, int gqa,
                                                     int rep, long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) uint4 kl[2][SWP_KB * 8];
  __shared__ __attribute__((aligned(16))) uint4 vl[2][SWP_KB * 8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tix = blockIdx.x;
  const int seq = qblocks[4 * tix], q0 = qblocks[4 * tix + 1], head = qblocks[4 * tix + 2];
  if (seq < 0) return;                       // padding entry of the XCD-interleaved order (whole block)
  const int s0 = cu[seq], S = cu[seq + 1] - s0;
  const int kvh = head / rep;
  const bf16_t* qbase = qkvg + (size_t)s0 * ld + head * 64;
  const bf16_t* gbase = qkvg + (size_t)s0 * ld + d_model + head * 64;
  const bf16_t* kbase = qkvg + (size_t)s0 * ld + 2 * d_model + kvh * 64;
  const bf16_t* vbase = kbase + gqa;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query r][16 ks + 8h + 0..7]; rows past the end are clamped, never stored
  const int qrow = q0 + wave * 32 + r;
  const int qrc = qrow < S ? qrow : S - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrc * ld + ks * 16 + h * 8);
  // a use of q ahead of the DMA statements: the compiler's wait for these loads (it cannot count the DMA of the asm statements,
  // so it waits for vmcnt(0)) lands here and not in front of the first MFMA
  asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));

  // K / V staging by LDS-DMA as in k_attn_bf16: instruction i of wave w covers tile rows 8 (2w + i) .. + 7, lane >> 3 the row,
  // lane & 7 the 16-byte LDS chunk; the XOR swizzles are applied on the global side (K chunk c holds global chunk
  // c ^ ((row >> 1) & 7), V chunk c holds c ^ (((row >> 1) & 1) << 2)).  Rows past the sequence end re-fetch its last row.
  const uint32_t kl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&kl[0][0];
  const uint32_t vl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&vl[0][0];
  const int drow0 = wave * 16 + (lane >> 3), drow1 = drow0 + 8;
  const int kc0 = ((lane & 7) ^ ((drow0 >> 1) & 7)) * 8, kc1 = ((lane & 7) ^ ((drow1 >> 1) & 7)) * 8;
  const int vc0 = ((lane & 7) ^ (((drow0 >> 1) & 1) << 2)) * 8, vc1 = ((lane & 7) ^ (((drow1 >> 1) & 1) << 2)) * 8;
  const uint32_t dK0 = (uint32_t)(drow0 * ld + kc0) * 2u, dK1 = (uint32_t)(drow1 * ld + kc1) * 2u;
  const uint32_t dV0 = (uint32_t)(drow0 * ld + vc0) * 2u, dV1 = (uint32_t)(drow1 * ld + vc1) * 2u;
#define SWP_DMA16(voff_, base_, dst_)                                                                            \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
  // one operand tile (two DMA instructions per wave): tile kt_ of operand base_ into ring slot (kt_ & 1); the tile is a SCALAR base,
  // the lane's share of it two constant offsets; only a sequence's last, partial tile computes clamped rows on the vector unit
#define SWP_DMA_TILE(base_, lds_, kt_, d0_, d1_, c0_, c1_)                                                       \
  do {                                                                                                           \
    const int key0__ = (kt_) * SWP_KB;                                                                           \
    const bf16_t* b__ = (base_) + (size_t)key0__ * ld;                                                           \
    const uint32_t dst__ = (lds_) + ((kt_) & 1) * (SWP_KB * 128) + wave * 2048;                                  \
    if (key0__ + SWP_KB <= S) {                                                                                  \
      unsigned keep__;                                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"         \
                   "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"             \
                   : "=&s"(keep__) : "v"(d0_), "v"(d1_), "s"(b__), "s"(dst__), "s"(dst__ + 1024) : "memory");       \
    } else {                                                                                                     \
      const int lim__ = S - 1 - key0__;                                                                          \
      const int g0__ = drow0 < lim__ ? drow0 : lim__, g1__ = drow1 < lim__ ? drow1 : lim__;                      \
      SWP_DMA16((uint32_t)(g0__ * ld + (c0_)) * 2u, b__, dst__);                                                 \
      SWP_DMA16((uint32_t)(g1__ * ld + (c1_)) * 2u, b__, dst__ + 1024);                                          \
    }                                                                                                            \
  } while (0)
#define SWP_DMA_K(kt_) SWP_DMA_TILE(kbase, kl_lds, kt_, dK0, dK1, kc0, kc1)
#define SWP_DMA_V(kt_) SWP_DMA_TILE(vbase, vl_lds, kt_, dV0, dV1, vc0, vc1)

  // ---- lane-constant LDS byte offsets (all per-tile variation is the slot base plus an immediate) ----
  // K fragment (A operand of S^T): key row 32 j + r, 16-byte chunk (2 ks + h) ^ ((row >> 1) & 7)
  const int ksw = (r >> 1) & 7;
  const char* const kbase_lds = reinterpret_cast<const char*>(&kl[0][0]);
  const char* const vbase_lds = reinterpret_cast<const char*>(&vl[0][0]);
  const char* const ka0 = kbase_lds + r * 128 + (((0 * 2 + h) ^ ksw) << 4);
  const char* const ka1 = kbase_lds + r * 128 + (((1 * 2 + h) ^ ksw) << 4);
  const char* const ka2 = kbase_lds + r * 128 + (((2 * 2 + h) ^ ksw) << 4);
  const char* const ka3 = kbase_lds + r * 128 + (((3 * 2 + h) ^ ksw) << 4);
  // V^T fragment via ds_read_b64_tr_b16: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p + 3 of its block; rows
  // 4h + tq (+ 16 sp + 32 j), 64-byte half (dt ^ ((row >> 1) & 1))
  const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
  const int vsw = (tq >> 1) & 1;
  const int vlane = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
  const char* const va0 = vbase_lds + vlane + (vsw ? 64 : 0);     // dt = 0
  const char* const va1 = vbase_lds + vlane + (vsw ? 0 : 64);     // dt = 1
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + SWP_KB - 1) / SWP_KB;
  f32x16 o0 = zero16, o1 = zero16;          // O^T tiles: head dims 0-31 / 32-63 x the wave's 32 queries
  f32x16 negm = zero16;                     // start vector of the score chains: -m of the lane's query in every element
  float m_run = 0.f, l_run = 0.f;
  f32x16 sA0, sA1, sB0, sB1;                // scores of two tiles in flight: keys 0-31 / 32-63 of the tile x 32 queries
  bf16x8 kf00, kf01, kf02, kf03, kf10, kf11, kf12, kf13;     // K fragments [key half][ks] of the NEXT score block
  bf16x8 vf00, vf01, vf02, vf03, vf10, vf11, vf12, vf13;     // V^T fragments [dt][k step] of the NEXT PV block
  bf16x8 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;             // P fragments (k steps 0..3) of the two tiles
  float sum_a, sum_b;                                         // the pending tile's two row-sum chains

#define SWP_LOADK(slot_)                                                                                         \
  do {                                                                                                           \
    const int so__ = (slot_) * (SWP_KB * 128);                                                                   \
    kf00 = *reinterpret_cast<const bf16x8*>(ka0 + so__);                                                         \
    kf01 = *reinterpret_cast<const bf16x8*>(ka1 + so__);                                                         \
    kf02 = *reinterpret_cast<const bf16x8*>(ka2 + so__);                                                         \
    kf03 = *reinterpret_cast<const bf16x8*>(ka3 + so__);                                                         \
    kf10 = *reinterpret_cast<const bf16x8*>(ka0 + so__ + 4096);                                                  \
    kf11 = *reinterpret_cast<const bf16x8*>(ka1 + so__ + 4096);                                                  \
    kf12 = *reinterpret_cast<const bf16x8*>(ka2 + so__ + 4096);                                                  \
    kf13 = *reinterpret_cast<const bf16x8*>(ka3 + so__ + 4096);                                                  \
  } while (0)
#define SWP_VFRAG(va_, OFF_)                                                                                     \
  ({                                                                                                             \
    const bf16x4 lo__ = swp_read_tr16((va_) + (OFF_)), hi__ = swp_read_tr16((va_) + (OFF_) + 1024);              \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                            \
  })
#define SWP_LOADV(slot_)                                                                                         \
  do {                                                                                                           \
    const int so__ = (slot_) * (SWP_KB * 128);                                                                   \
    vf00 = SWP_VFRAG(va0, so__);        vf10 = SWP_VFRAG(va1, so__);                                             \
    vf01 = SWP_VFRAG(va0, so__ + 2048); vf11 = SWP_VFRAG(va1, so__ + 2048);                                      \
    vf02 = SWP_VFRAG(va0, so__ + 4096); vf12 = SWP_VFRAG(va1, so__ + 4096);                                      \
    vf03 = SWP_VFRAG(va0, so__ + 6144); vf13 = SWP_VFRAG(va1, so__ + 6144);                                      \
  } while (0)
#define SWP_PACK8(p_)                                                                                            \
  ((bf16x8){(bf16_t)p_[0], (bf16_t)p_[1], (bf16_t)p_[2], (bf16_t)p_[3], (bf16_t)p_[4], (bf16_t)p_[5], (bf16_t)p_[6], (bf16_t)p_[7]})
  // exp2 of the 16 scores of one key half -> two P fragments; the row-sum chains take every p.  FIRST_: the chains start here.
#define SWP_EXP16(s_, f0_, f1_, FIRST_)                                                                          \
  do {                                                                                                           \
    float p__[16];                                                                                               \
    _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) p__[e__] = __builtin_amdgcn_exp2f(s_[e__]);             \
    if (FIRST_) { sum_a = p__[0]; sum_b = p__[1]; } else { sum_a += p__[0]; sum_b += p__[1]; }                   \
    _Pragma("unroll") for (int e__ = 2; e__ < 16; e__ += 2) { sum_a += p__[e__]; sum_b += p__[e__ + 1]; }        \
    f0_ = SWP_PACK8((&p__[0]));                                                                                  \
    f1_ = SWP_PACK8((&p__[8]));                                                                                  \
  } while (0)
  // row maximum of a tile's 64 scores over both lane halves of a query
#define SWP_ROWMAX(c0_, c1_)                                                                                     \
  ({                                                                                                             \
    float a__ = fmaxf(fmaxf(c0_[0], c0_[1]), c0_[2]), b__ = fmaxf(fmaxf(c1_[0], c1_[1]), c1_[2]);                \
    _Pragma("unroll") for (int e__ = 3; e__ < 15; e__ += 2) {                                                    \
      a__ = fmaxf(fmaxf(a__, c0_[e__]), c0_[e__ + 1]);                                                           \
      b__ = fmaxf(fmaxf(b__, c1_[e__]), c1_[e__ + 1]);                                                           \
    }                                                                                                            \
    const float m2__ = fmaxf(fmaxf(a__, c0_[15]), fmaxf(b__, c1_[15]));                                          \
    const auto sw__ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2__), __builtin_bit_cast(unsigned, m2__), false, false); \
    const unsigned u0__ = sw__[0], u1__ = sw__[1];     /* not __builtin_bit_cast(float, sw__[1]): hipcc 7.2 reads element 0 for both */ \
    fmaxf(__uint_as_float(u0__), __uint_as_float(u1__));                                                         \
  })
  // keys past the end of the sequence (only in the last tile, only when S is not a multiple of 64)
#define SWP_MASK(c0_, c1_, kt_)                                                                                  \
  do {                                                                                                           \
    if ((kt_) == nkt - 1 && nkt * SWP_KB > S) {                                                                  \
      int h4__ = 4 * h;                                                                                          \
      asm volatile("" : "+v"(h4__));      /* computed here, once per sequence: not 32 loop-invariant lane masks in SGPRs */ \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) {                                                     \
        const int key__ = (kt_) * SWP_KB + (e__ & 3) + 8 * (e__ >> 2) + h4__;                                    \
        if (key__ >= S) c0_[e__] = -INFINITY;                                                                    \
        if (key__ + 32 >= S) c1_[e__] = -INFINITY;                                                               \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
  // the 8 MFMAs of S = K Q^T - m from the K fragments in registers: two chains of four, each starting from the -m vector
#define SWP_SCORES(d0_, d1_, c_)                                                                                 \
  do {                                                                                                           \
    d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf00, qf[0], c_, 0, 0, 0);                                     \
    d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf01, qf[1], d0_, 0, 0, 0);                                    \
    d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf02, qf[2], d0_, 0, 0, 0);                                    \
    d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf03, qf[3], d0_, 0, 0, 0);                                    \
    d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf10, qf[0], c_, 0, 0, 0);                                     \
    d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf11, qf[1], d1_, 0, 0, 0);                                    \
    d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf12, qf[2], d1_, 0, 0, 0);                                    \
    d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf13, qf[3], d1_, 0, 0, 0);                                    \
  } while (0)
  // the 8 MFMAs of O^T += V^T P^T from the V fragments in registers
#define SWP_PV(f0_, f1_, f2_, f3_)                                                                               \
  do {                                                                                                           \
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf00, f0_, o0, 0, 0, 0);                                        \
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf10, f0_, o1, 0, 0, 0);                                        \
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf01, f1_, o0, 0, 0, 0);                                        \
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf11, f1_, o1, 0, 0, 0);                                        \
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf02, f2_, o0, 0, 0, 0);                                        \
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf12, f2_, o1, 0, 0, 0);                                        \
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf03, f3_, o0, 0, 0, 0);                                        \
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf13, f3_, o1, 0, 0, 0);                                        \
  } while (0)
  // per MFMA gap: the MFMA, its share of the phase's LDS reads, then 2 transcendentals and 3 plain vector instructions
#define SWP_GAPS(NDS_)                                                                                           \
  do {                                                                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 8; ++i__) {                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, NDS_, 0);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                                                         \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                         \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)

  // One iteration for tile t_.  Entering: cur_ = raw scores S(t_) (relative to the reference), keys 0-31 of it already
  // exponentiated into pc0_, pc1_ (sum chains running); K fragments of tile t_+1 in kf.  Leaving: the same for t_ + 1 with
  // (cur_, pc_) <-> (nxt_, pn_) swapped.
#define SWP_ITER(t_, cur0_, cur1_, nxt0_, nxt1_, pc0_, pc1_, pc2_, pc3_, pn0_, pn1_)                             \
  do {                                                                                                           \
    const int tt__ = (t_);                                                                                       \
    /* vmcnt(0): this wave's shares of V(t), K(t+2) (issued an iteration ago); lgkmcnt(0): its K(t+1) fragment reads of Y(t-1) */ \
    /* have RETURNED (they are consumed right below anyway) - the slot they came from is refilled behind the barrier */   \
    __builtin_amdgcn_s_waitcnt(0x0070);                                                                          \
    SWP_STAMP(0);                                                                                                \
    __builtin_amdgcn_s_barrier();         /* ... everybody's; all waves are done with V(t-1) and K(t+1) */        \
    SWP_STAMP(1);                                                                                                \
    if (tt__ + 1 < nkt) SWP_DMA_V(tt__ + 1);                                                                     \
    if (tt__ + 3 < nkt) SWP_DMA_K(tt__ + 3);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    SWP_STAMP(2);                                                                                                \
    /* ---- X(t): S(t+1) | keys 32-63 of tile t | V(t) fragments ---- */                                         \
    SWP_LOADV(tt__ & 1);                                                                                         \
    SWP_SCORES(nxt0_, nxt1_, negm);                                                                              \
    SWP_EXP16(cur1_, pc2_, pc3_, false);                                                                         \
    /* pinned to this block: their consumers sit behind the check, where the compiler would sink the packing */  \
    asm volatile("" : "+v"(pc2_), "+v"(pc3_), "+v"(sum_a), "+v"(sum_b));                                         \
    SWP_GAPS(2);                                                                                                 \
    SWP_STAMP(3);                                                                                                \
    /* ---- check(t): was the reference good enough for tile t? ---- */                                          \
    float ps__ = sum_a + sum_b;                                                                                  \
    if (__builtin_amdgcn_ballot_w64(!(ps__ <= SWP_BIG)) != 0ull) {                                               \
      /* rare: move the reference to the tile's true maximum.  O and l hold tiles < t against the old reference, S(t) and */ \
      /* S(t+1) are relative to it, the start vector carries it: all shift by d; then the tile's P again. */      \
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");     /* S(t+1)'s last MFMA may still be writing */ \
      const float d__ = fmaxf(SWP_ROWMAX(cur0_, cur1_), 0.f);                                                    \
      const float alpha__ = __builtin_amdgcn_exp2f(-d__);                                                        \
      l_run *= alpha__;                                                                                          \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) {                                                     \
        o0[e__] *= alpha__; o1[e__] *= alpha__;                                                                  \
        cur0_[e__] -= d__; cur1_[e__] -= d__; nxt0_[e__] -= d__; nxt1_[e__] -= d__;                              \
      }                                                                                                          \
      m_run += d__;                                                                                              \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) negm[e__] = -m_run;                                   \
      SWP_EXP16(cur0_, pc0_, pc1_, true);                                                                        \
      SWP_EXP16(cur1_, pc2_, pc3_, false);                                                                       \
      ps__ = sum_a + sum_b;                                                                                      \
    }                                                                                                            \
    l_run += ps__;                                                                                               \
    if (tt__ + 1 == nkt - 1 && nkt * SWP_KB > S) {                                                               \
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");                                                \
      SWP_MASK(nxt0_, nxt1_, tt__ + 1);                                                                          \
    }                                                                                                            \
    SWP_STAMP(4);                                                                                                \
    /* ---- Y(t): O += V(t)^T P(t)^T | keys 0-31 of tile t+1 | K(t+2) fragments ---- */                          \
    SWP_LOADK(tt__ & 1);                                                                                         \
    SWP_PV(pc0_, pc1_, pc2_, pc3_);                                                                              \
    SWP_EXP16(nxt0_, pn0_, pn1_, true);                                                                          \
    asm volatile("" : "+v"(pn0_), "+v"(pn1_), "+v"(sum_a), "+v"(sum_b));     /* consumed an iteration later */   \
    SWP_GAPS(1);                                                                                                 \
    SWP_STAMP(5);                                                                                                \
  } while (0)

  // ---- prologue: K(0), V(0), K(1) in flight; S(0), its row maximum = the reference; K(2) behind it; keys 0-31 of tile 0 ----
  SWP_DMA_K(0);
  SWP_DMA_V(0);
  if (nkt > 1) SWP_DMA_K(1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_s_barrier();
  SWP_LOADK(0);
  SWP_SCORES(sA0, sA1, zero16);
  SWP_MASK(sA0, sA1, 0);
  {
    m_run = SWP_ROWMAX(sA0, sA1);
#pragma unroll
    for (int e = 0; e < 16; ++e) { sA0[e] -= m_run; sA1[e] -= m_run; negm[e] = -m_run; }
  }
  __builtin_amdgcn_s_barrier();             // every wave has read K(0): its slot takes K(2)
  if (nkt > 2) SWP_DMA_K(2);
  SWP_LOADK(1);                             // K(1) landed with K(0) (one wait above); garbage when nkt == 1 (unused)
  SWP_EXP16(sA0, pa0, pa1, true);
  SWP_STAMP_DECL;
  SWP_STAMP_START();

  // ---- key loop, two tiles per trip (the score sets swap roles) ----
  int t = 0;
  for (;;) {
    SWP_ITER(t, sA0, sA1, sB0, sB1, pa0, pa1, pa2, pa3, pb0, pb1);
    if (++t == nkt) break;
    SWP_ITER(t, sB0, sB1, sA0, sA1, pb0, pb1, pb2, pb3, pa0, pa1);
    if (++t == nkt) break;
  }

#ifdef SWP_STAMPS
  if (stamps && blockIdx.x % 37 == 0 && lane == 0) {      // [0..5] loop segments of the wave, [6] its number of key tiles
    long long* dst = stamps + ((size_t)(blockIdx.x / 37) * 4 + wave) * 8;
    for (int i = 0; i < 6; ++i) dst[i] = (long long)st_acc__[i];
    dst[6] = (long long)nkt;
    dst[7] = 0;
  }
#endif

  // ---- normalise, gate, store (as k_attn_bf16): lane holds O[query r][32 dt + 8 g + 4 h + 0..3] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = __builtin_amdgcn_rcpf(l_tot);
  {
    bf16_t* orow = out + (size_t)(s0 + qrc) * ldo + head * 64;
    const bf16_t* grow = gbase + (size_t)qrc * ld;
    const bool store = qrow < S;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        uint2 pk[2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * gp + gg;
          const int d0 = dt * 32 + 8 * g + 4 * h;
          const f32x16& oa = dt ? o1 : o0;
          f32x4 v = {oa[4 * g] * inv_l, oa[4 * g + 1] * inv_l, oa[4 * g + 2] * inv_l, oa[4 * g + 3] * inv_l};
          if (GATE) {
            const f32x4 gt = Vec4<bf16_t>::load(grow + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt[e] * -1.44269504088896340736f));
          }
          const bf16x4 b4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          pk[gg] = __builtin_bit_cast(uint2, b4);
        }
        // lanes 32..63 of pk[0] <-> lanes 0..31 of pk[1]: one 16-byte store per lane and pair of 4-feature groups
        const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const uint4 o16 = {sx[0], sy[0], sx[1], sy[1]};
        if (store) *reinterpret_cast<uint4*>(orow + dt * 32 + 16 * gp + 8 * h) = o16;
      }
  }
}

// Launcher: called by ttvk_attention for bf16 tables of full items with pre-scaled q and no tape outputs.
int ttvk_attention_swp(const void* qkvg, int ld, void* out, int ldo, const int* cu_seqlens, const int* qblocks, int n_qblocks,
                       int q_heads, int kv_heads, int gate_mul, hipStream_t s) {
  const int d_model = q_heads * 64, gqa = kv_heads * 64, rep = q_heads / kv_heads;
  dim3 grid(n_qblocks);
  if (gate_mul)
    hipLaunchKernelGGL((k_attn_swp<true>), grid, dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, cu_seqlens, qblocks, d_model,
                       gqa, rep, g_ttv_stamps);
  else
    hipLaunchKernelGGL((k_attn_swp<false>), grid, dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, cu_seqlens, qblocks, d_model,
                       gqa, rep, g_ttv_stamps);
  TTV_CHECK_LAUNCH("attention_swp");
  return TTV_OK;
}

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
//   phase Y(t):  8 MFMAs  O += V(t)^T P(t)^T    |  exp2 / pack / row sums of keys  0-31 of tile t+1 |   8 K(t+2) fragment reads
//
// per MFMA exactly 2 v_exp_f32 + 1 v_cvt_pk_bf16_f32 + 2 v_add_f32 (issue cost 8 + 2 x 8.7 + 3 x 3 = 34 cycles against the MFMA's
// 32: tools/ubench/valu_rates.hip), pinned per gap with sched_group_barrier.  What makes the mix that thin:
//   * q arrives multiplied by scale * log2(e) (ttv_layer_weights.qkv_q_prescaled): a score leaves the matrix pipe as the exponent;
//   * NO softmax reference in the loop: p = exp2(score) as it stands.  Any reference gives the same quotient, bf16 P keeps its 8
//     significant bits at any magnitude and O / l are fp32, so all a reference does is keep the exponents inside fp32's range - and
//     whether they were is read off the row sums at the END: every row of the block must have 2^-60 < l < 2^60 (then no p overflowed
//     and whatever underflowed was below 2^-66 of its row's sum).  Otherwise - a block-uniform branch, rare by construction: some
//     exponent beyond ~60 - the block runs again through a plain exact loop (running maximum per tile, rescale on every change) that
//     is correct for any score range.  No row maximum, no subtraction, no start vector, no check inside the loop;
//   * row sums are plain v_add_f32 chains (v_pk_add_f32 beside MFMAs costs more than the two adds it replaces).
// Registers: S(t) 32 + S(t+1) 32 + P 16..24 + O 32 + K fragments 32 + V fragments 32 + q 16 + addresses: two waves per SIMD
// (launch bounds 256 x 2), 48 KB LDS per block (two 3-slot rings).
//
// LDS hand-over: ONE barrier per tile at the top of iteration t.  Behind it every wave has finished X(t-1) / Y(t-1), i.e. its reads of
// V(t-1) and K(t+1): V(t+2) is issued into V(t-1)'s slot (between the halves of X(t), under its MFMAs) and K(t+4) into K(t+1)'s
// (between the halves of Y(t)).  Every tile issues exactly four DMA instructions per wave (tile indices past the end re-fetch the last
// tile into a slot nobody reads), so the wait in front of the barrier is the constant `s_waitcnt vmcnt(4)`: the four newest stay in
// flight, V(t+1) and K(t+3) - issued two tiles earlier, first read in X(t+1) / Y(t+1) - have landed.
// (First version of this file, same round: reference = maximum of the first tile carried as the MFMAs' C operand, a check of the row
// sums per tile with an in-loop rare path, DMA behind the barrier, 2-slot rings: 63.6 us against k_attn_bf16's 61.6 - the stamps
// (tools/swp_stamps.py) showed 1 034 of a tile's 1 859 cycles in the two MFMA phases and the rest in DMA issue 333, check 257, wait +
// barrier 234, with the two co-resident blocks of a CU running their phases in step.)
#include <stdlib.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

#define SWP_KB 64
#define SWP_NS 3          // ring slots per operand
#ifndef SWP_OCC
#define SWP_OCC 3         // waves per SIMD the register allocation aims at (blocks per CU)
#endif

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
__global__ __launch_bounds__(256, SWP_OCC) void k_attn_swp(const bf16_t* __restrict__ qkvg, int ld, bf16_t* __restrict__ out, int ldo,
                                                     const int* __restrict__ cu, const int* __restrict__ qblocks, int d_model, int gqa,
                                                     int rep, long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) uint4 kl[SWP_NS][SWP_KB * 8];
  __shared__ __attribute__((aligned(16))) uint4 vl[SWP_NS][SWP_KB * 8];

#ifdef SWP_STAMPS
  unsigned long long st_entry__;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_entry__)::"memory");
#endif
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
  const int nkt = (S + SWP_KB - 1) / SWP_KB;

  // K / V staging by LDS-DMA as in k_attn_bf16: lane >> 3 picks the row of an 8-row piece,
  // lane & 7 the 16-byte LDS chunk; the XOR swizzles are applied on the global side (K chunk c holds global chunk
  // c ^ ((row >> 1) & 7), V chunk c holds c ^ (((row >> 1) & 1) << 2)).  Rows past the sequence end re-fetch its last row.
  const uint32_t kl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&kl[0][0];
  const uint32_t vl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&vl[0][0];
  // Wave w stages tile rows 8 w + (lane >> 3) and that + 32 of K and of V.  Rows 32 apart have the same chunk swizzle, so ONE per-lane
  // offset serves both instructions of an operand (the 32 rows go into the scalar base): two VGPRs live through the loop, not four -
  // four were two too many for the 168 of three waves per SIMD (spilled and reloaded in the loop behind a vmcnt(0) that drained the DMA).
  const int drow = wave * 8 + (lane >> 3);
  const int kc = ((lane & 7) ^ ((drow >> 1) & 7)) * 8;
  const int vc = ((lane & 7) ^ (((drow >> 1) & 1) << 2)) * 8;
  const uint32_t dK = (uint32_t)(drow * ld + kc) * 2u;
  const uint32_t dV = (uint32_t)(drow * ld + vc) * 2u;
#define SWP_DMA16(voff_, base_, dst_)                                                                            \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
  // One operand tile = two DMA instructions per wave into ring slot slot_; tiles past the sequence's end are not issued (SWP_TOP
  // counts accordingly).  The tile is a SCALAR base, the lane's share of it one constant offset; only a sequence's last, partial
  // tile computes clamped rows on the vector unit.
#define SWP_DMA_TILE(base_, lds_, kt_, slot_, d_, c_)                                                            \
  do {                                                                                                           \
    const int key0__ = (kt_) * SWP_KB;                                                                           \
    if (key0__ >= S) break;                                                                                      \
    const bf16_t* b__ = (base_) + (size_t)key0__ * ld;                                                           \
    const uint32_t dst__ = (lds_) + (slot_) * (SWP_KB * 128) + wave * 1024;                                      \
    if (key0__ + SWP_KB <= S) {                                                                                  \
      const bf16_t* b32__ = b__ + (size_t)32 * ld;                                                               \
      unsigned keep__;                                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"         \
                   "s_add_u32 m0, %4, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"      \
                   : "=&s"(keep__) : "v"(d_), "s"(b__), "s"(b32__), "s"(dst__) : "memory", "scc");                   \
    } else {                                                                                                     \
      const int lim__ = S - 1 - key0__;                                                                          \
      const int g0__ = drow < lim__ ? drow : lim__, g1__ = drow + 32 < lim__ ? drow + 32 : lim__;                \
      SWP_DMA16((uint32_t)(g0__ * ld + (c_)) * 2u, b__, dst__);                                                  \
      SWP_DMA16((uint32_t)(g1__ * ld + (c_)) * 2u, b__, dst__ + 4096);                                           \
    }                                                                                                            \
  } while (0)
#define SWP_DMA_K(kt_, slot_) SWP_DMA_TILE(kbase, kl_lds, kt_, slot_, dK, kc)
#define SWP_DMA_V(kt_, slot_) SWP_DMA_TILE(vbase, vl_lds, kt_, slot_, dV, vc)

  // ---- prologue, part 1: K(0), V(0), K(1), K(2), V(1) in flight before anything is waited for; then Q ----
  SWP_DMA_K(0, 0);
  SWP_DMA_V(0, 0);
  SWP_DMA_K(1, 1);
  SWP_DMA_K(2, 2);
  SWP_DMA_V(1, 1);
  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query r][16 ks + 8h + 0..7]; rows past the end are clamped, never stored
  const int qrow = q0 + wave * 32 + r;
  const int qrc = qrow < S ? qrow : S - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrc * ld + ks * 16 + h * 8);

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

  f32x16 o0 = zero16, o1 = zero16;          // O^T tiles: head dims 0-31 / 32-63 x the wave's 32 queries
  float l_run = 0.f;
  f32x16 sA0, sA1, sB0, sB1;                // scores of two tiles in flight: keys 0-31 / 32-63 of the tile x 32 queries
  bf16x8 kf00, kf01, kf02, kf03, kf10, kf11, kf12, kf13;     // K fragments [key half][ks] of the NEXT score block
  bf16x8 vf00, vf01, vf02, vf03, vf10, vf11, vf12, vf13;     // V^T fragments [dt][k step] of the NEXT PV block
  bf16x8 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;             // P fragments (k steps 0..3) of the two tiles
  float sum_a, sum_b;                                         // the pending tile's two row-sum chains

#define SWP_LOADK_H(J_, so_)                                                                                     \
  do {                                                                                                           \
    if ((J_) == 0) {                                                                                             \
      kf00 = *reinterpret_cast<const bf16x8*>(ka0 + (so_));                                                      \
      kf01 = *reinterpret_cast<const bf16x8*>(ka1 + (so_));                                                      \
      kf02 = *reinterpret_cast<const bf16x8*>(ka2 + (so_));                                                      \
      kf03 = *reinterpret_cast<const bf16x8*>(ka3 + (so_));                                                      \
    } else {                                                                                                     \
      kf10 = *reinterpret_cast<const bf16x8*>(ka0 + (so_) + 4096);                                               \
      kf11 = *reinterpret_cast<const bf16x8*>(ka1 + (so_) + 4096);                                               \
      kf12 = *reinterpret_cast<const bf16x8*>(ka2 + (so_) + 4096);                                               \
      kf13 = *reinterpret_cast<const bf16x8*>(ka3 + (so_) + 4096);                                               \
    }                                                                                                            \
  } while (0)
#define SWP_VFRAG(va_, OFF_)                                                                                     \
  ({                                                                                                             \
    const bf16x4 lo__ = swp_read_tr16((va_) + (OFF_)), hi__ = swp_read_tr16((va_) + (OFF_) + 1024);              \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                            \
  })
  // V fragments of k steps 2 J_, 2 J_ + 1 (keys 32 J_ .. 32 J_ + 31 of the tile)
#define SWP_LOADV_H(J_, so_)                                                                                     \
  do {                                                                                                           \
    if ((J_) == 0) {                                                                                             \
      vf00 = SWP_VFRAG(va0, (so_));        vf10 = SWP_VFRAG(va1, (so_));                                         \
      vf01 = SWP_VFRAG(va0, (so_) + 2048); vf11 = SWP_VFRAG(va1, (so_) + 2048);                                  \
    } else {                                                                                                     \
      vf02 = SWP_VFRAG(va0, (so_) + 4096); vf12 = SWP_VFRAG(va1, (so_) + 4096);                                  \
      vf03 = SWP_VFRAG(va0, (so_) + 6144); vf13 = SWP_VFRAG(va1, (so_) + 6144);                                  \
    }                                                                                                            \
  } while (0)
  // exp2 of 8 scores (elements 8 Q_ .. 8 Q_ + 7 of a score set) -> one P fragment; the row-sum chains take every p (plain adds:
  // v_pk_add_f32 beside MFMAs costs more than the two adds it replaces).  FIRST_: the chains start here.
#define SWP_EXP8(s_, Q_, f_, FIRST_)                                                                             \
  do {                                                                                                           \
    float p__[8];                                                                                                \
    _Pragma("unroll") for (int e__ = 0; e__ < 8; ++e__) p__[e__] = __builtin_amdgcn_exp2f(s_[8 * (Q_) + e__]);   \
    if (FIRST_) { sum_a = p__[0]; sum_b = p__[1]; } else { sum_a += p__[0]; sum_b += p__[1]; }                   \
    _Pragma("unroll") for (int e__ = 2; e__ < 8; e__ += 2) { sum_a += p__[e__]; sum_b += p__[e__ + 1]; }         \
    f_ = (bf16x8){(bf16_t)p__[0], (bf16_t)p__[1], (bf16_t)p__[2], (bf16_t)p__[3], (bf16_t)p__[4], (bf16_t)p__[5],  \
                  (bf16_t)p__[6], (bf16_t)p__[7]};                                                               \
    /* pinned to this block: the consumers sit a phase or more later, where the compiler would sink the arithmetic */ \
    asm volatile("" : "+v"(f_), "+v"(sum_a), "+v"(sum_b));                                                       \
  } while (0)
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
  // four MFMAs of S = K Q^T for one key half, from the K fragments in registers
#define SWP_SCORES_H(d_, k0_, k1_, k2_, k3_)                                                                     \
  do {                                                                                                           \
    d_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0_, qf[0], zero16, 0, 0, 0);                                   \
    d_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1_, qf[1], d_, 0, 0, 0);                                       \
    d_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2_, qf[2], d_, 0, 0, 0);                                       \
    d_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k3_, qf[3], d_, 0, 0, 0);                                       \
  } while (0)
  // four MFMAs of O^T += V^T P^T for two k steps
#define SWP_PV_H(va_, vb_, vc_, vd_, f0_, f1_)                                                                   \
  do {                                                                                                           \
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_, f0_, o0, 0, 0, 0);                                         \
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb_, f0_, o1, 0, 0, 0);                                         \
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc_, f1_, o0, 0, 0, 0);                                         \
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vd_, f1_, o1, 0, 0, 0);                                         \
  } while (0)
  // four MFMA gaps: the MFMA, its share of the block's LDS reads, then 2 transcendentals and 3 plain vector instructions
#define SWP_GAPS4(NDS_)                                                                                          \
  do {                                                                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, NDS_, 0);                                                      \
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                                                         \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                         \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)

  // Top of iteration t_: V(t_) and K(t_+2) (issued two tiles ago) must have landed.  The only younger DMA instructions are those of
  // V(t_+1) and K(t_+3) (issued in X / Y of the previous iteration, two each, where those tiles exist): they may stay in flight.  The
  // wave's K(t_+1) fragment reads of Y(t_-1) have RETURNED (lgkmcnt(0)): the slot they came from is refilled behind the barrier.
#define SWP_TOP(t_)                                                                                              \
  do {                                                                                                           \
    if ((t_) + 3 < nkt) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                              \
    else if ((t_) + 1 < nkt) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                         \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                             \
    SWP_STAMP(0);                                                                                                \
    asm volatile("s_barrier" ::: "memory");     /* ... everybody's; all waves are done with V(t-1) and K(t+1) */  \
    SWP_STAMP(1);                                                                                                \
  } while (0)

  // One iteration for tile t_.  Entering: cur_ = scores S(t_) (exponents), keys 0-31 of it already exponentiated into pc0_, pc1_
  // (sum chains running); K fragments of tile t_+1 in kf.  Leaving: the same for t_ + 1 with (cur_, pc_) <-> (nxt_, pn_) swapped.
  // so0_ / so1_ / so2_: byte offsets of the ring slots of tiles t_, t_+1, t_+2 (rotated by the caller).
#define SWP_ITER(t_, so0_, so1_, so2_, cur1_, nxt0_, nxt1_, pc0_, pc1_, pc2_, pc3_, pn0_, pn1_)                  \
  do {                                                                                                           \
    const int tt__ = (t_);                                                                                       \
    SWP_TOP(tt__);                                                                                               \
    /* ---- X(t): S(t+1) | keys 32-63 of tile t | V(t) fragments; V(t+2) goes out between its halves ---- */      \
    SWP_LOADV_H(0, so0_);                                                                                        \
    SWP_SCORES_H(nxt0_, kf00, kf01, kf02, kf03);                                                                 \
    SWP_EXP8(cur1_, 0, pc2_, false);                                                                             \
    SWP_GAPS4(2);                                                                                                \
    SWP_DMA_V(tt__ + 2, (so2_) >> 13);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    SWP_LOADV_H(1, so0_);                                                                                        \
    SWP_SCORES_H(nxt1_, kf10, kf11, kf12, kf13);                                                                 \
    SWP_EXP8(cur1_, 1, pc3_, false);                                                                             \
    SWP_GAPS4(2);                                                                                                \
    SWP_STAMP(2);                                                                                                \
    l_run += sum_a + sum_b;                                                                                      \
    if (tt__ + 1 == nkt - 1 && nkt * SWP_KB > S) {                                                               \
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");     /* S(t+1)'s last MFMA may still be writing */ \
      SWP_MASK(nxt0_, nxt1_, tt__ + 1);                                                                          \
    }                                                                                                            \
    SWP_STAMP(3);                                                                                                \
    /* ---- Y(t): O += V(t)^T P(t)^T | keys 0-31 of tile t+1 | K(t+2) fragments; K(t+4) goes out between its halves ---- */ \
    SWP_LOADK_H(0, so2_);                                                                                        \
    SWP_PV_H(vf00, vf10, vf01, vf11, pc0_, pc1_);                                                                \
    SWP_EXP8(nxt0_, 0, pn0_, true);                                                                              \
    SWP_GAPS4(1);                                                                                                \
    SWP_DMA_K(tt__ + 4, (so1_) >> 13);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    SWP_LOADK_H(1, so2_);                                                                                        \
    SWP_PV_H(vf02, vf12, vf03, vf13, pc2_, pc3_);                                                                \
    SWP_EXP8(nxt0_, 1, pn1_, false);                                                                             \
    SWP_GAPS4(1);                                                                                                \
    SWP_STAMP(4);                                                                                                \
  } while (0)

  // ---- prologue, part 2: S(0); K(3) behind it; K(1) fragments; keys 0-31 of tile 0 ----
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): Q and this wave's shares of the five tiles
  __syncthreads();
  SWP_LOADK_H(0, 0);
  SWP_LOADK_H(1, 0);
  SWP_SCORES_H(sA0, kf00, kf01, kf02, kf03);
  SWP_SCORES_H(sA1, kf10, kf11, kf12, kf13);
  SWP_MASK(sA0, sA1, 0);
  __syncthreads();                          // every wave has read K(0): its slot takes K(3)
  SWP_DMA_K(3, 0);
  SWP_LOADK_H(0, SWP_KB * 128);             // K(1) landed with K(0) (one wait above); stale bytes when nkt == 1 (scores unused)
  SWP_LOADK_H(1, SWP_KB * 128);
  SWP_EXP8(sA0, 0, pa0, true);
  SWP_EXP8(sA0, 1, pa1, false);
  SWP_STAMP_DECL;
  SWP_STAMP_START();
#ifdef SWP_STAMPS
  const unsigned long long st_loop_start__ = st_prev__;
  unsigned long long st_real0__;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_real0__)::"memory");
#endif

  // The sequence's last tile: nothing to prepare for a tile behind it - keys 32-63 of the tile, then its eight PV MFMAs
#define SWP_LAST(t_, so0_, cur1_, pc0_, pc1_, pc2_, pc3_)                                                        \
  do {                                                                                                           \
    SWP_TOP(t_);                                                                                                 \
    SWP_LOADV_H(0, so0_);                                                                                        \
    SWP_PV_H(vf00, vf10, vf01, vf11, pc0_, pc1_);                                                                \
    SWP_EXP8(cur1_, 0, pc2_, false);                                                                             \
    SWP_GAPS4(2);                                                                                                \
    SWP_LOADV_H(1, so0_);                                                                                        \
    SWP_EXP8(cur1_, 1, pc3_, false);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    SWP_STAMP(2);                                                                                                \
    l_run += sum_a + sum_b;                                                                                      \
    SWP_PV_H(vf02, vf12, vf03, vf13, pc2_, pc3_);                                                                \
    SWP_STAMP(4);                                                                                                \
  } while (0)

  // ---- key loop, two tiles per trip (the score sets swap roles); ring slot offsets rotate in scalar registers.  The last tile runs
  // behind the loop from the "A" registers (an exit in the other role moves its 24 live registers over: once per block) ----
  {
    int t = 0;
    int so0 = 0, so1 = SWP_KB * 128, so2 = 2 * SWP_KB * 128;
    for (;;) {
      if (t == nkt - 1) break;
      SWP_ITER(t, so0, so1, so2, sA1, sB0, sB1, pa0, pa1, pa2, pa3, pb0, pb1);
      ++t;
      if (t == nkt - 1) {
        sA1 = sB1; pa0 = pb0; pa1 = pb1; so0 = so1;
        break;
      }
      SWP_ITER(t, so1, so2, so0, sB1, sA0, sA1, pb0, pb1, pb2, pb3, pa0, pa1);
      ++t;
      const int so = so0;       // two tiles on: slots (t, t+1, t+2) = old (t+2, t+3 = t, t+4 = t+1)
      so0 = so2;
      so2 = so1;
      so1 = so;
    }
    SWP_LAST(t, so0, sA1, pa0, pa1, pa2, pa3);
  }
#ifdef SWP_STAMPS
  const unsigned long long st_loop_end__ = st_prev__;
  unsigned long long st_real1__;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_real1__)::"memory");
  st_acc__[5] = st_real1__ - st_real0__;       // the loop in ticks of the constant 100 MHz clock: shader clock = loop cycles / this x 100 MHz
#endif

  // ---- the row sums say whether the reference-free exponents were in range; otherwise the whole block again, exactly ----
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const bool out_of_range = !(l_tot > 8.6736174e-19f && l_tot < 1.1529215e18f);       // [2^-60, 2^60]; NaN fails both
  if (__syncthreads_or(out_of_range)) {
    // Exact online softmax (running maximum per tile, rescale on every change), one tile at a time through ring slot 0: any score
    // range the fp32 exponent can express.  Rare by construction (|exponent| beyond ~60 somewhere in the block's rows).
    float m_run = -INFINITY;
    l_run = 0.f;
    o0 = zero16;
    o1 = zero16;
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();                                  // every wave is done with the previous tile
      SWP_DMA_K(kt, 0);
      SWP_DMA_V(kt, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      SWP_LOADK_H(0, 0);
      SWP_LOADK_H(1, 0);
      SWP_SCORES_H(sA0, kf00, kf01, kf02, kf03);
      SWP_SCORES_H(sA1, kf10, kf11, kf12, kf13);
      SWP_MASK(sA0, sA1, kt);
      float mx = fmaxf(sA0[0], sA1[0]);
#pragma unroll
      for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(sA0[e], sA1[e]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);             // finite from the first tile on (it holds at least one key)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        sA0[e] = __builtin_amdgcn_exp2f(sA0[e] - m_new);
        sA1[e] = __builtin_amdgcn_exp2f(sA1[e] - m_new);
        ps += sA0[e] + sA1[e];
        o0[e] *= alpha;
        o1[e] *= alpha;
      }
      l_run = l_run * alpha + ps;
      m_run = m_new;
#define SWP_PACKS(c_, Q_)                                                                                         \
  ((bf16x8){(bf16_t)c_[8 * Q_ + 0], (bf16_t)c_[8 * Q_ + 1], (bf16_t)c_[8 * Q_ + 2], (bf16_t)c_[8 * Q_ + 3], (bf16_t)c_[8 * Q_ + 4], \
            (bf16_t)c_[8 * Q_ + 5], (bf16_t)c_[8 * Q_ + 6], (bf16_t)c_[8 * Q_ + 7]})
      pa0 = SWP_PACKS(sA0, 0); pa1 = SWP_PACKS(sA0, 1); pa2 = SWP_PACKS(sA1, 0); pa3 = SWP_PACKS(sA1, 1);
#undef SWP_PACKS
      SWP_LOADV_H(0, 0);
      SWP_LOADV_H(1, 0);
      SWP_PV_H(vf00, vf10, vf01, vf11, pa0, pa1);
      SWP_PV_H(vf02, vf12, vf03, vf13, pa2, pa3);
    }
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }

  // ---- normalise, gate, store (as k_attn_bf16): lane holds O[query r][32 dt + 8 g + 4 h + 0..3] ----
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
#ifdef SWP_STAMPS
  if (stamps && blockIdx.x % 37 == 0 && lane == 0) {      // [0..4] loop segments of the wave, [6] entry -> loop, [7] loop end -> stores acknowledged
    unsigned long long st_end__;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_end__)::"memory");
    long long* dst = stamps + ((size_t)(blockIdx.x / 37) * 4 + wave) * 8;
    for (int i = 0; i < 6; ++i) dst[i] = (long long)st_acc__[i];
    dst[6] = (long long)(st_loop_start__ - st_entry__);
    dst[7] = (long long)(st_end__ - st_loop_end__);
  }
#endif
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

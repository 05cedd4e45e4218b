// Variable-length, non-causal GQA attention, bf16, head_dim 64: the 64-query-rows-per-wave kernel (round 3).
// Replaces flash_attn_varlen_func at reference model/base/transformer.py:100 and the sigmoid gate at :103 for inference towers
// whose q columns arrive pre-scaled (ttv_layer_weights.qkv_q_prescaled); same arithmetic as k_attn_bf16<.., PRE = true> (ttv_attn.hip).
//
// Structure:
//   * one workgroup = 4 waves, TWO workgroups per CU (two waves per SIMD, 256 registers each: 96 accumulation registers owned by
//     the asm statements below + <= 160 VGPRs).  A wave owns 64 query rows (two 32-row tiles A and B) of one q-head; the four
//     waves of a block work on the same (sequence, kv-head) - any q-head of that kv-head, any 64-row slice - and share every
//     K / V tile (work table built by the host, plan.attention_table64).
//   * every K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) read from LDS feeds TWO MFMAs (tile A and tile B):
//     half the LDS reads, LDS-DMA instructions, scalar bookkeeping and barriers per MFMA of the 32-rows-per-wave kernel.
//   * K / V tiles of 64 keys arrive by LDS-DMA (global_load_lds_dwordx4, XOR swizzle on the source side; K three tiles ahead in a
//     four-slot ring, V two ahead in a three-slot ring), ONE raw s_barrier per tile behind a counted s_waitcnt vmcnt(4).  The
//     arithmetic runs in 32-key SUB-STEPS (two per tile): one sub-step's score tiles, P fragments and K / V fragments are 96
//     registers instead of 192 - what lets two waves share a SIMD.  (The first build of this kernel kept whole 64-key steps in
//     registers at ONE wave per SIMD, the structure cdna_hip_programming.md documents at 50 % of peak for head_dim 128: correct,
//     and 1.6x SLOWER than k_attn_bf16 at head_dim 64 - 3 300 cycles per step for 1 024 cycles of MFMA.  With twice the softmax
//     per MFMA a lone wave is bound by its own instruction issue, ~8 cycles per instruction over ~400 instructions per step;
//     knock-out builds: no in-loop DMA -14 %, no fragment reads -12 %, no row maximum -9 %, no exp2 -13 %, all four -56 %.)
//   * the two tiles run half a sub-step apart, so that the matrix pipe and the vector unit always have independent work from
//     the same wave; sub-step v = 32 keys:
//           Ya:  PV_B(v-1)  beside  row maximum of S_A(v)                              [+ one LDS-DMA instruction]
//           --   rare: move tile A's softmax reference (wave-uniform branch; O_B's MFMAs may be in flight, O_A's are not)
//           Yb:  S_B(v)     beside  exp2 / row sum / bf16 pack of S_A(v) -> P_A(v)     [+ V(v) fragment reads]
//           Xa:  PV_A(v)    beside  row maximum of S_B(v)                              [+ K(v+1) fragment reads, one LDS-DMA instruction]
//           --   rare: move tile B's reference
//           Xb:  S_A(v+1)   beside  exp2 / row sum / pack of S_B(v) -> P_B(v)
//     A reference shift of one tile never meets pending P.V MFMAs of the same tile (cdna_hip_programming.md T13 hazard): when
//     tile A's reference moves, PV_A(v-1) is complete and PV_A(v) not yet issued, and its P is computed afterwards.
//   * softmax as in k_attn_bf16<PRE>: the q columns carry scale * log2(e), the score accumulators start from -m (running
//     reference per query row), so a score goes from the accumulator straight into v_exp_f32; the reference moves only when a
//     score exceeds it by more than `defer_thr` (log2 units).
#include <stdlib.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

#define KB 64
// clobber lists of the asm-owned accumulation registers (see W64_MFMA_O*)
#define W64_ACL_A0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define W64_ACL_A1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define W64_ACL_B0 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define W64_ACL_Q "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", \
                  "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define W64_ACL_B1 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"

#define W64_QWRITE(qf_, n0_, n1_, n2_, n3_)                                                                      \
  do {                                                                                                           \
    const uint4 q4__ = __builtin_bit_cast(uint4, qf_);                                                           \
    asm volatile("v_accvgpr_write_b32 a" #n0_ ", %0\n\tv_accvgpr_write_b32 a" #n1_ ", %1\n\tv_accvgpr_write_b32 a" #n2_ ", %2\n\t"   \
                 "v_accvgpr_write_b32 a" #n3_ ", %3\n\ts_nop 1" :: "v"(q4__.x), "v"(q4__.y), "v"(q4__.z), "v"(q4__.w) : W64_ACL_Q);  \
  } while (0)

// the lane id, recomputed where it is needed in rarely taken paths and after the key loop: a lane-derived value that is live across
// the loop costs a VGPR there (the kernel runs at 160 VGPRs + 96 accumulation registers; a spill inside the loop would also put a
// compiler-counted scratch load - and its s_waitcnt vmcnt(0) - between the hand-counted LDS-DMA waits)
__device__ __forceinline__ int w64_fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 lds64_read_tr16(const char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds_ptr));
}

// Diagnostic build only (-DATTN64_STAMPS): s_memtime sums per loop segment, written by every 37th block.
#ifdef ATTN64_STAMPS
#define W64_STAMP_DECL unsigned long long st_prev__ = 0, st_acc__[6] = {0, 0, 0, 0, 0, 0}
#define W64_STAMP_START()                                                                              \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev__)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
#define W64_STAMP(seg_)                                                                                \
  do {                                                                                                 \
    unsigned long long t__;                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    st_acc__[seg_] += t__ - st_prev__;                                                                 \
    st_prev__ = t__;                                                                                   \
  } while (0)
#else
#define W64_STAMP_DECL
#define W64_STAMP_START()
#define W64_STAMP(seg_)
#endif

// knock-out switches of diagnostic builds (tools/attn64_knockout.sh): what does the loop cost WITHOUT the in-loop DMA, the LDS fragment
// reads, the row maximum + shift check, the exp2?  Results are garbage then; the product build has all of them at 0.
#ifndef W64_KO_DMA
#define W64_KO_DMA 0
#endif
#ifndef W64_KO_LDS
#define W64_KO_LDS 0
#endif
#ifndef W64_KO_MAX
#define W64_KO_MAX 0
#endif
#ifndef W64_KO_EXP
#define W64_KO_EXP 0
#endif

// work table: int32 [n_items, 8] = (sequence, kv-head, wave 0..3: q-head | (first query row / 64) << 8, or -1 = idle wave,
// first packed row of the sequence, its length);
// sequence < 0 = padding entry of the XCD-interleaved order
template <bool GATE>
__global__ __launch_bounds__(256, 2) void k_attn_w64(const bf16_t* __restrict__ qkvg, int ld, bf16_t* __restrict__ out, int ldo,
                                                     const int* __restrict__ cu, const int* __restrict__ items, int d_model, int gqa, int rep,
                                                     float defer_thr, long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(1024))) uint4 kl[4][KB * 8];     // K ring: tiles u .. u+3
  __shared__ __attribute__((aligned(1024))) uint4 vl[3][KB * 8];     // V ring: tiles u .. u+2
#ifdef ATTN64_STAMPS
  unsigned long long st_entry__;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_entry__)::"memory");
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int* item = items + 8 * blockIdx.x;
  const int seq = item[0];
  if (seq < 0) return;
  const int kvh = item[1];
  const int wd = item[2 + wave];
  const bool live = wd >= 0;                       // an idle wave shadows rows of its block's kv-head (staging share, barriers), stores nothing
  const int head = live ? (wd & 0xff) : kvh * rep;
  const int q0 = live ? (wd >> 8) * 64 : 0;
  const int s0 = item[6], S = item[7];             // = cu[seq], cu[seq + 1] - cu[seq]: in the table, one dependent load less
  const bf16_t* qbase = qkvg + (size_t)s0 * ld + head * 64;
  const bf16_t* gbase = qkvg + (size_t)s0 * ld + d_model + head * 64;
  const bf16_t* kbase = qkvg + (size_t)s0 * ld + 2 * d_model + kvh * 64;
  const bf16_t* vbase = kbase + gqa;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query][16 ks + 8 h + 0..7] of its row in tile A and in tile B
  const int qrowA = q0 + r, qrowB = q0 + 32 + r;
  const int qrcA = qrowA < S ? qrowA : S - 1, qrcB = qrowB < S ? qrowB : S - 1;
  bf16x8 qfA[4], qfB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qfA[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrcA * ld + ks * 16 + h * 8);
    qfB[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrcB * ld + ks * 16 + h * 8);
  }
  // DMA shares (as k_attn_pipe): wave w stages tile rows 8 w + (lane >> 3) and that + 32 of K and of V; rows 32 apart have the same
  // chunk swizzle (K: (row >> 1) & 7, V: ((row >> 1) & 1) << 2), so one per-lane offset serves both instructions of an operand
  const uint32_t kl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&kl[0][0];
  const uint32_t vl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&vl[0][0];
  const int drow = wave * 8 + (lane >> 3);
  const int kc = ((lane & 7) ^ ((drow >> 1) & 7)) * 8;
  const int vc = ((lane & 7) ^ (((drow >> 1) & 1) << 2)) * 8;
  const uint32_t dK = (uint32_t)(drow * ld + kc) * 2u;
  const uint32_t dV = (uint32_t)(drow * ld + vc) * 2u;
  const int nkt = (S + KB - 1) / KB;
#define W64_DMA(voff_, base_, dst_)                                                                              \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
  // one operand tile (two DMA instructions per wave, always): tile index clamped to the last one, rows past the end clamped
#define W64_TILE(base_, lds_, kt_, soff_, c_, d_)                                                                \
  do {                                                                                                           \
    const int ktc__ = (kt_) < nkt ? (kt_) : nkt - 1;                                                             \
    const int key0__ = ktc__ * KB;                                                                               \
    const bf16_t* b__ = (base_) + (size_t)key0__ * ld;                                                           \
    const uint32_t dst__ = (lds_) + (soff_) + wave * 1024;                                                       \
    if (key0__ + KB <= S) {                                                                                      \
      W64_DMA(d_, b__, dst__);                                                                                   \
      W64_DMA(d_, b__ + (size_t)32 * ld, dst__ + 4096);                                                          \
    } else {                                                                                                     \
      const int lim__ = S - 1 - key0__;                                                                          \
      const int g0__ = drow < lim__ ? drow : lim__, g1__ = drow + 32 < lim__ ? drow + 32 : lim__;                \
      W64_DMA((uint32_t)(g0__ * ld + (c_)) * 2u, b__, dst__);                                                    \
      W64_DMA((uint32_t)(g1__ * ld + (c_)) * 2u, b__, dst__ + 4096);                                             \
    }                                                                                                            \
  } while (0)
  // ONE DMA instruction: half half_ (tile rows 32 half_ .. + 31; literal 0 / 1) of an operand tile.  steady_: the tile is a full
  // tile before the sequence's last one and run_ its running 64-bit base (no clamping, no index arithmetic).  The four
  // instructions of a step are issued one per region: back to back (right behind the barrier, in all four waves at once) each
  // of them held its wave for ~110 cycles - in-kernel stamps of the first build: 475 cycles per iteration.
#define W64_HALF(steady_, run_, base_, lds_, kt_, soff_, ISK_, d_, half_)                                        \
  do {                                                                                                           \
    if (W64_KO_DMA) break;                                                                                       \
    if (steady_) {                                                                                               \
      W64_DMA(d_, (run_) + (size_t)(32 * (half_)) * ld, (lds_) + (soff_) + wave * 1024 + 4096 * (half_));        \
    } else {                                                                                                     \
      const int ktc__ = (kt_) < nkt ? (kt_) : nkt - 1;                                                           \
      const int key0__ = ktc__ * KB;                                                                             \
      const bf16_t* b__ = (base_) + (size_t)key0__ * ld;                                                         \
      const uint32_t dst__ = (lds_) + (soff_) + wave * 1024 + 4096 * (half_);                                    \
      if (key0__ + KB <= S) {                                                                                    \
        W64_DMA(d_, b__ + (size_t)(32 * (half_)) * ld, dst__);                                                   \
      } else {                                                                                                   \
        const int ln__ = w64_fresh_lane();                                                                       \
        const int dr__ = wave * 8 + (ln__ >> 3);                                                                 \
        const int c__ = (ISK_) ? ((ln__ & 7) ^ ((dr__ >> 1) & 7)) * 8 : ((ln__ & 7) ^ (((dr__ >> 1) & 1) << 2)) * 8; \
        const int lim__ = S - 1 - key0__;                                                                        \
        const int g__ = dr__ + 32 * (half_) < lim__ ? dr__ + 32 * (half_) : lim__;                               \
        W64_DMA((uint32_t)(g__ * ld + c__) * 2u, b__, dst__);                                                    \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)

  // lane-constant LDS byte offsets (layout of k_attn_bf16), ONE per operand; the ring slot is a scalar byte offset
  // K fragment (A operand of S^T): key row 32 kb + r, 16-byte chunk (2 ks + h) ^ ((row >> 1) & 7) - the k-step only toggles
  // address bits 5, 6: offset(ks) = offset(0) ^ 32 ks
  const char* const kbase_lds = reinterpret_cast<const char*>(&kl[0][0]);
  const char* const vbase_lds = reinterpret_cast<const char*>(&vl[0][0]);
  int koff0, voff0;
  {
    const int r = lane & 31, h = lane >> 5;
    koff0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
    // V^T fragment via ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its block; head dims
    // 0..31 at the 64-byte half (row >> 1) & 1 of the row, head dims 32..63 at the other: offset(dt = 1) = offset(dt = 0) ^ 64
    const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
    voff0 = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8 + (((tq >> 1) & 1) ? 64 : 0);
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};

  // ---- state ----
  // O^T accumulators: a[0:63], see W64_MFMA_O*
#define W64_AZ1(n_) "v_accvgpr_write_b32 a" #n_ ", 0\n\t"
#define W64_AZERO16(n0_, n1_, n2_, n3_, n4_, n5_, n6_, n7_, n8_, n9_, n10_, n11_, n12_, n13_, n14_, n15_, cl_)                \
  asm volatile(W64_AZ1(n0_) W64_AZ1(n1_) W64_AZ1(n2_) W64_AZ1(n3_) W64_AZ1(n4_) W64_AZ1(n5_) W64_AZ1(n6_) W64_AZ1(n7_) W64_AZ1(n8_)  \
               W64_AZ1(n9_) W64_AZ1(n10_) W64_AZ1(n11_) W64_AZ1(n12_) W64_AZ1(n13_) W64_AZ1(n14_) W64_AZ1(n15_) "s_nop 1" ::: cl_)
  W64_AZERO16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, W64_ACL_A0);
  W64_AZERO16(16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, W64_ACL_A1);
  W64_AZERO16(32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, W64_ACL_B0);
  W64_AZERO16(48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, W64_ACL_B1);
  f32x16 sA, sB = zero16;                                          // score tiles of the current 32-key sub-step: tile A, tile B
  f32x16 negmA = zero16, negmB = zero16;                           // start vectors of the score accumulators: -m per query (lane)
  float mA = 0.f, mB = 0.f, lA = 0.f, lB = 0.f;
  bf16x8 pA0, pA1;                                                  // P of the sub-step as B fragments, k-step sp
  bf16x8 pB0 = zero8, pB1 = zero8;
  bf16x8 kf0, kf1, kf2, kf3;                                        // K fragments [ks] of the sub-step S is computed from next
  bf16x8 vf00 = zero8, vf01 = zero8, vf10 = zero8, vf11 = zero8;   // V^T fragments [dt][sp] of the current sub-step

  // Every MFMA is an asm statement: the compiler, left to choose where accumulation registers are available, accumulates the score
  // tiles in AGPRs too and then moves every score through v_accvgpr_read before the softmax (576 copies per iteration in the first
  // build), or keeps O in VGPRs across the loop and copies it to AGPRs and back around every MFMA.  Here the O^T accumulators and
  // the Q fragments are asm-owned accumulation registers, the score tiles VGPR operands ("v").  An asm statement is opaque to the
  // scheduler, so the interleave is written out: a SLOT is one MFMA followed by a chunk of vector work, fenced with
  // sched_barrier(0), and every chunk's results are pinned by an empty asm (pure arithmetic is otherwise sunk to its first use).
  // Hazards the compiler does not pad for asm (cdna_hip_programming.md 5.7): the score tiles are read by vector instructions a
  // region later (>= 12 instructions behind their last MFMA, see the slot lists); P fragments, K / V fragments and the -m vectors
  // are written at least a slot before the MFMA that reads them; O is read only in the rare shift (its last MFMA is a region
  // back) and in the epilogue (behind an explicit s_nop).
#define W64_FENCE() __builtin_amdgcn_sched_barrier(0)
  // The O^T accumulators are asm-owned accumulation registers: tile A a[0:31] (head dims 0-31 | 32-63), tile B a[32:63].  (As C++
  // variables with "+a" operands the compiler kept them in VGPRs across the loop and copied all 64 to AGPRs and back every
  // iteration.)  Every statement that touches them lists them as clobbers, which also makes the kernel descriptor allocate them.
#define W64_MFMA_OA0(a_, b_) do { asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a_), "v"(b_) : W64_ACL_A0); W64_FENCE(); } while (0)
#define W64_MFMA_OA1(a_, b_) do { asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(a_), "v"(b_) : W64_ACL_A1); W64_FENCE(); } while (0)
#define W64_MFMA_OB0(a_, b_) do { asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(a_), "v"(b_) : W64_ACL_B0); W64_FENCE(); } while (0)
#define W64_MFMA_OB1(a_, b_) do { asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(a_), "v"(b_) : W64_ACL_B1); W64_FENCE(); } while (0)
  // a_n *= alpha for 16 consecutive accumulation registers starting at literal n0_ (rare path; one temporary, the vector unit
  // interlocks the read -> multiply -> write chain)
#define W64_ASC1(n_) "v_accvgpr_read_b32 %0, a" #n_ "\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a" #n_ ", %0\n\t"
#define W64_ASCALE16(alpha_, n0_, n1_, n2_, n3_, n4_, n5_, n6_, n7_, n8_, n9_, n10_, n11_, n12_, n13_, n14_, n15_, cl_)        \
  do {                                                                                                           \
    float t__;                                                                                                   \
    asm volatile(W64_ASC1(n0_) W64_ASC1(n1_) W64_ASC1(n2_) W64_ASC1(n3_) W64_ASC1(n4_) W64_ASC1(n5_) W64_ASC1(n6_) W64_ASC1(n7_)   \
                 W64_ASC1(n8_) W64_ASC1(n9_) W64_ASC1(n10_) W64_ASC1(n11_) W64_ASC1(n12_) W64_ASC1(n13_) W64_ASC1(n14_) W64_ASC1(n15_) "s_nop 1" \
                 : "=&v"(t__) : "v"(alpha_) : cl_);                                                              \
  } while (0)
#define W64_OSCALE_A(alpha_)                                                                                     \
  do {                                                                                                           \
    W64_ASCALE16(alpha_, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, W64_ACL_A0);                      \
    W64_ASCALE16(alpha_, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, W64_ACL_A1);            \
  } while (0)
#define W64_OSCALE_B(alpha_)                                                                                     \
  do {                                                                                                           \
    W64_ASCALE16(alpha_, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, W64_ACL_B0);            \
    W64_ASCALE16(alpha_, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, W64_ACL_B1);            \
  } while (0)
  // four accumulation registers n0_ .. n0_ + 3 (literals) -> f32x4
#define W64_AREAD4(dst_, n0_, n1_, n2_, n3_)                                                                     \
  do {                                                                                                           \
    float x0__, x1__, x2__, x3__;                                                                                \
    asm volatile("v_accvgpr_read_b32 %0, a" #n0_ "\n\tv_accvgpr_read_b32 %1, a" #n1_ "\n\tv_accvgpr_read_b32 %2, a" #n2_       \
                 "\n\tv_accvgpr_read_b32 %3, a" #n3_ : "=v"(x0__), "=v"(x1__), "=v"(x2__), "=v"(x3__));             \
    dst_ = (f32x4){x0__, x1__, x2__, x3__};                                                                      \
  } while (0)
  // the Q fragments (B operand of every score MFMA, loaded once per block) live in accumulation registers as well: tile A
  // a[64:79], tile B a[80:95], k-step ks at + 4 ks - an MFMA takes its A / B operands from either file.  32 VGPRs less: the -m
  // start vectors then stay in VGPRs (with Q in VGPRs the compiler parked one of them in AGPRs and copied it back every iteration)
#define W64_MFMA_S(s_, a_, qlo_, qhi_)                                                                           \
  do {                                                                                                           \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[" #qlo_ ":" #qhi_ "], %0" : "+v"(s_) : "v"(a_) : W64_ACL_Q); \
    W64_FENCE();                                                                                                 \
  } while (0)
  // first MFMA of a score chain: starts from the tile's -m vector and leaves it intact (fresh destination range)
#define W64_MFMA_S0(s_, a_, qlo_, qhi_, negm_)                                                                   \
  do {                                                                                                           \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[" #qlo_ ":" #qhi_ "], %2" : "=&v"(s_) : "v"(a_), "v"(negm_) : W64_ACL_Q); \
    W64_FENCE();                                                                                                 \
  } while (0)
  // K fragments [ks] of key rows 32 kb .. + 31 of a ring slot: off_ = slot byte offset + 4096 kb
#define W64_LOADK(off_)                                                                                          \
  do {                                                                                                           \
    if (W64_KO_LDS) break;                                                                                       \
    const int t__ = koff0 + (off_);                                                                              \
    kf0 = *reinterpret_cast<const bf16x8*>(kbase_lds + t__);                                                     \
    kf1 = *reinterpret_cast<const bf16x8*>(kbase_lds + (t__ ^ 32));                                              \
    kf2 = *reinterpret_cast<const bf16x8*>(kbase_lds + (t__ ^ 64));                                              \
    kf3 = *reinterpret_cast<const bf16x8*>(kbase_lds + (t__ ^ 96));                                              \
  } while (0)
#define W64_VFRAG(va_, OFF_)                                                                                     \
  ({                                                                                                             \
    const bf16x4 lo__ = lds64_read_tr16((va_) + (OFF_)), hi__ = lds64_read_tr16((va_) + (OFF_) + 1024);          \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                            \
  })
  // V^T fragments [dt][sp] of key rows 32 kb .. + 31 of a ring slot (off_ = slot byte offset + 4096 kb), half i_ (literal 0 / 1 = sp)
#define W64_LOADV(off_, i_)                                                                                      \
  do {                                                                                                           \
    if (W64_KO_LDS) break;                                                                                       \
    const int t__ = voff0 + (off_);                                                                              \
    if ((i_) == 0) { vf00 = W64_VFRAG(vbase_lds + t__, 0);    vf10 = W64_VFRAG(vbase_lds + (t__ ^ 64), 0); }     \
    if ((i_) == 1) { vf01 = W64_VFRAG(vbase_lds + t__, 2048); vf11 = W64_VFRAG(vbase_lds + (t__ ^ 64), 2048); }  \
  } while (0)
  // keys past the end of the sequence (last tile of a sequence whose length is not a multiple of 64)
#define W64_MASK(s_, kt_, kb_)                                                                                   \
  do {                                                                                                           \
    if ((kt_) == nkt - 1 && nkt * KB > S) {                                                                      \
      const int lim__ = S - (kt_) * KB - 32 * (kb_) - 4 * (w64_fresh_lane() >> 5);   /* opaque: no precomputed lane masks in the loop's scalar registers */ \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__)                                                       \
        if ((e__ & 3) + 8 * (e__ >> 2) >= lim__) s_[e__] = -INFINITY;                                            \
    }                                                                                                            \
  } while (0)
  // row maximum of the 32 scores of a query in this sub-step: two chains over 8 registers, then the combine over both lane halves
#define W64_MAXCHAIN(dst_, s_, b_)                                                                               \
  do {                                                                                                           \
    if (W64_KO_MAX) { dst_ = s_[(b_)]; break; }                                                                  \
    float a__ = fmaxf(fmaxf(s_[(b_) + 0], s_[(b_) + 1]), s_[(b_) + 2]);                                          \
    a__ = fmaxf(fmaxf(a__, s_[(b_) + 3]), s_[(b_) + 4]);                                                         \
    a__ = fmaxf(fmaxf(a__, s_[(b_) + 5]), s_[(b_) + 6]);                                                         \
    dst_ = fmaxf(a__, s_[(b_) + 7]);                                                                             \
  } while (0)
#define W64_MAXFIN(c0_, c1_)                                                                                     \
  ({                                                                                                             \
    const float m2__ = fmaxf(c0_, c1_);                                                                          \
    const auto sw__ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2__), __builtin_bit_cast(unsigned, m2__), false, false); \
    const unsigned u0__ = sw__[0], u1__ = sw__[1];     /* not __builtin_bit_cast(float, sw__[1]): hipcc 7.2 reads element 0 for both */ \
    fmaxf(__uint_as_float(u0__), __uint_as_float(u1__));                                                         \
  })
  // the tile's scores are relative to its running reference (they started from -m).  The reference moves only when some score of
  // the tile exceeds it by more than defer_thr (first tile: always, the reference is the placeholder 0): scores, row sum, O and
  // the start vector are shifted (see k_attn_bf16)
#define W64_SHIFT(first_, mx_, s_, OSCALE_, l_, m_, negm_)                                                       \
  do {                                                                                                           \
    if (W64_KO_MAX) break;                                                                                       \
    if ((first_) || __builtin_amdgcn_ballot_w64((mx_) > defer_thr) != 0ull) {                                    \
      const float d__ = (first_) ? (mx_) : fmaxf((mx_), 0.f);                                                    \
      if (!(first_)) {                                            /* nothing accumulated yet in the first sub-step */ \
        const float alpha__ = __builtin_amdgcn_exp2f(-d__);                                                      \
        l_ *= alpha__;                                                                                           \
        OSCALE_(alpha__);                                                                                        \
      }                                                                                                          \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) s_[e__] -= d__;                                       \
      m_ += d__;                                                                                                 \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) negm_[e__] = -m_;                                     \
      asm volatile("s_nop 1" ::: "memory");                                                                      \
    }                                                                                                            \
  } while (0)
  // softmax of one tile's sub-step in four chunks of four scores (chunk c_: registers 4 c_ .. + 3): E = exp2 in place, A = add to the
  // four partial row sums, C = pack to bf16 into the P fragment sp = c_ >> 1.  Every chunk pins its results (empty asm): pure
  // arithmetic is otherwise sunk into the block of its first use, past every fence.
#define W64_E(s_, c_)                                                                                            \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) {                                                        \
      if (W64_KO_EXP) continue;                                                                                  \
      s_[4 * (c_) + j__] = __builtin_amdgcn_exp2f(s_[4 * (c_) + j__]);                                           \
    }                                                                                                            \
    asm volatile("" : "+v"(s_));                                                                                 \
  } while (0)
#define W64_A(s_, c_)                                                                                            \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) {                                                        \
      if ((c_) == 0) ps__[j__] = s_[j__];                                                                        \
      else ps__[j__] += s_[4 * (c_) + j__];                                                                      \
    }                                                                                                            \
    asm volatile("" : "+v"(ps__[0]), "+v"(ps__[1]), "+v"(ps__[2]), "+v"(ps__[3]));                               \
  } while (0)
#define W64_C(s_, c_, pf_)                                                                                       \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) pf_[4 * ((c_) & 1) + j__] = (bf16_t)s_[4 * (c_) + j__];  \
    asm volatile("" : "+v"(pf_));                                                                                \
  } while (0)

  // ---- prologue: K(0) | V(0), K(1) | K(2), V(1) in flight (10 DMA instructions per wave); S_A of the first sub-step ----
  W64_TILE(kbase, kl_lds, 0, 0, kc, dK);
  W64_TILE(vbase, vl_lds, 0, 0, vc, dV);
  W64_TILE(kbase, kl_lds, 1, KB * 128, kc, dK);
  W64_TILE(kbase, kl_lds, 2, 2 * KB * 128, kc, dK);
  W64_TILE(vbase, vl_lds, 1, KB * 128, vc, dV);
  // Q -> accumulation registers (see W64_MFMA_S), BEHIND the issue of the first K / V tiles: the q rows and the tiles then travel
  // together (the compiler's wait for the q loads is vmcnt(0): it cannot count the DMA of the asm statements, so it also waits
  // for the five tiles - K(0) is needed next anyway)
  W64_QWRITE(qfA[0], 64, 65, 66, 67); W64_QWRITE(qfA[1], 68, 69, 70, 71); W64_QWRITE(qfA[2], 72, 73, 74, 75); W64_QWRITE(qfA[3], 76, 77, 78, 79);
  W64_QWRITE(qfB[0], 80, 81, 82, 83); W64_QWRITE(qfB[1], 84, 85, 86, 87); W64_QWRITE(qfB[2], 88, 89, 90, 91); W64_QWRITE(qfB[3], 92, 93, 94, 95);
  const bf16_t* kdma = kbase + (size_t)3 * KB * ld;      // K(u+3), V(u+2) for u = 0
  const bf16_t* vdma = vbase + (size_t)2 * KB * ld;
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // K(0): this wave's share
  __builtin_amdgcn_s_barrier();
  W64_LOADK(0);
  W64_FENCE();
  W64_MFMA_S0(sA, kf0, 64, 67, negmA);
  W64_MFMA_S(sA, kf1, 68, 71);
  W64_MFMA_S(sA, kf2, 72, 75);
  W64_MFMA_S(sA, kf3, 76, 79);
  W64_STAMP_DECL;
  W64_STAMP_START();
#ifdef ATTN64_STAMPS
  const unsigned long long st_loop_start__ = st_prev__;
#endif

  // One sub-step (32 keys: rows 32 KB_ .. + 31 of tile u; KB_ literal 0 / 1).  Entering: S_A(v) raw in sA, its K fragments are
  // spent; P_B(v-1) in pB and V(v-1) in vf; kf = K fragments of THIS sub-step (for S_B).  DMA_A_ / DMA_X_: the LDS-DMA
  // instruction issued in Ya / Xa.  koff_next_: LDS byte offset of the next sub-step's K rows, voff_: of this sub-step's V rows.
#define W64_SUB(KB_, first_, voff_, koff_next_, DMA_A_, DMA_X_, SEG_)                                            \
  do {                                                                                                           \
    float c0__, c1__;                                                                                            \
    /* ---- Ya: PV_B(v-1) | row maximum of S_A(v) */                                                             \
    W64_MASK(sA, u, KB_);                                                                                        \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OB0(vf00, pB0);                                                                                     \
    DMA_A_;                                                                                                      \
    W64_MAXCHAIN(c0__, sA, 0);                                                                                   \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OB1(vf10, pB0);                                                                                     \
    W64_MAXCHAIN(c1__, sA, 8);                                                                                   \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OB0(vf01, pB1);                                                                                     \
    const float mxA__ = W64_MAXFIN(c0__, c1__);                                                                  \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OB1(vf11, pB1);                                                                                     \
    W64_SHIFT(first_, mxA__, sA, W64_OSCALE_A, lA, mA, negmA);                                                   \
    W64_FENCE();                                                                                                 \
    W64_STAMP(SEG_);                                                                                             \
    /* ---- Yb: V(v) fragments; S_B(v) | exp2 / row sum / pack of S_A(v) */                                      \
    {                                                                                                            \
      float ps__[4];                                                                                             \
      W64_MFMA_S0(sB, kf0, 80, 83, negmB);                                                                       \
      W64_LOADV(voff_, 0);                                                                                       \
      W64_E(sA, 0);                                                                                              \
      W64_FENCE();                                                                                               \
      W64_MFMA_S(sB, kf1, 84, 87);                                                                               \
      W64_LOADV(voff_, 1);                                                                                       \
      W64_E(sA, 1); W64_A(sA, 0);                                                                                \
      W64_FENCE();                                                                                               \
      W64_MFMA_S(sB, kf2, 88, 91);                                                                               \
      W64_E(sA, 2); W64_A(sA, 1); W64_C(sA, 0, pA0);                                                             \
      W64_FENCE();                                                                                               \
      W64_MFMA_S(sB, kf3, 92, 95);                                                                               \
      W64_E(sA, 3); W64_A(sA, 2); W64_C(sA, 1, pA0);                                                             \
      W64_FENCE();                                                                                               \
      W64_A(sA, 3); W64_C(sA, 2, pA1);                                                                           \
      W64_C(sA, 3, pA1);                                                                                         \
      lA += (ps__[0] + ps__[1]) + (ps__[2] + ps__[3]);                                                           \
      asm volatile("" : "+v"(lA));                                                                               \
      W64_FENCE();                                                                                               \
    }                                                                                                            \
    W64_STAMP(SEG_ + 1);                                                                                         \
    W64_MASK(sB, u, KB_);                                                                                        \
    W64_FENCE();                                                                                                 \
    /* ---- Xa: PV_A(v); K(v+1) fragments | row maximum of S_B(v) */                                             \
    W64_MFMA_OA0(vf00, pA0);                                                                                     \
    DMA_X_;                                                                                                      \
    W64_LOADK(koff_next_);                                                                                       \
    W64_MAXCHAIN(c0__, sB, 0);                                                                                   \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OA1(vf10, pA0);                                                                                     \
    W64_MAXCHAIN(c1__, sB, 8);                                                                                   \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OA0(vf01, pA1);                                                                                     \
    const float mxB__ = W64_MAXFIN(c0__, c1__);                                                                  \
    W64_FENCE();                                                                                                 \
    W64_MFMA_OA1(vf11, pA1);                                                                                     \
    W64_SHIFT(first_, mxB__, sB, W64_OSCALE_B, lB, mB, negmB);                                                   \
    W64_FENCE();                                                                                                 \
    W64_STAMP(SEG_ + 2);                                                                                         \
    /* ---- Xb: S_A(v+1) | exp2 / row sum / pack of S_B(v) */                                                    \
    {                                                                                                            \
      float ps__[4];                                                                                             \
      W64_MFMA_S0(sA, kf0, 64, 67, negmA);      /* past the last tile: scores of the re-fetched tile, unused */  \
      W64_E(sB, 0);                                                                                              \
      W64_FENCE();                                                                                               \
      W64_MFMA_S(sA, kf1, 68, 71);                                                                               \
      W64_E(sB, 1); W64_A(sB, 0);                                                                                \
      W64_FENCE();                                                                                               \
      W64_MFMA_S(sA, kf2, 72, 75);                                                                               \
      W64_E(sB, 2); W64_A(sB, 1); W64_C(sB, 0, pB0);                                                             \
      W64_FENCE();                                                                                               \
      W64_MFMA_S(sA, kf3, 76, 79);                                                                               \
      W64_E(sB, 3); W64_A(sB, 2); W64_C(sB, 1, pB0);                                                             \
      W64_FENCE();                                                                                               \
      W64_A(sB, 3); W64_C(sB, 2, pB1);                                                                           \
      W64_C(sB, 3, pB1);                                                                                         \
      lB += (ps__[0] + ps__[1]) + (ps__[2] + ps__[3]);                                                           \
      asm volatile("" : "+v"(lB));                                                                               \
      W64_FENCE();                                                                                               \
    }                                                                                                            \
    W64_STAMP(SEG_ + 3);                                                                                         \
  } while (0)

  int ov0 = 0, ov1 = KB * 128, ov2 = 2 * KB * 128;       // V ring slot byte offsets of tiles u, u+1, u+2 (scalar registers)
  for (int u = 0; u < nkt; ++u) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // K(u+1), V(u) landed (own share); K(u+2), V(u+1) may fly
    __builtin_amdgcn_s_barrier();                        // ... for every wave; all are done reading K(u-1) and V(u-1) from LDS
    W64_STAMP(0);
    // K(u+3) -> K ring slot (u+3) & 3 (= that of K(u-1)), V(u+2) -> the slot of V(u-1): one DMA instruction per region
    const bool steady = u + 3 < nkt - 1;
    const int ok0 = (u & 3) * (KB * 128), ok1 = ((u + 1) & 3) * (KB * 128), ok3 = ((u + 3) & 3) * (KB * 128);
    W64_SUB(0, u == 0, ov0, ok0 + 4096, W64_HALF(steady, kdma, kbase, kl_lds, u + 3, ok3, true, dK, 0),
            W64_HALF(steady, kdma, kbase, kl_lds, u + 3, ok3, true, dK, 1), 1);
    W64_SUB(1, false, ov0 + 4096, ok1, W64_HALF(steady, vdma, vbase, vl_lds, u + 2, ov2, false, dV, 0),
            W64_HALF(steady, vdma, vbase, vl_lds, u + 2, ov2, false, dV, 1), 1);
    kdma += (size_t)KB * ld;
    vdma += (size_t)KB * ld;
    const int o = ov0;
    ov0 = ov1;
    ov1 = ov2;
    ov2 = o;
  }
  // ---- drain: PV_B of the last sub-step (its V fragments were read in its Yb) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the block's LDS is released
  W64_MFMA_OB0(vf00, pB0);
  W64_MFMA_OB1(vf10, pB0);
  W64_MFMA_OB0(vf01, pB1);
  W64_MFMA_OB1(vf11, pB1);
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the accumulators are read by vector instructions next (asm MFMAs are not padded)
  W64_FENCE();
#ifdef ATTN64_STAMPS
  const unsigned long long st_loop_end__ = st_prev__;
#endif

  // ---- normalise, gate, store: lane holds O[query r][32 dt + 8 g + 4 h + 0..3] of tile A and of tile B ----
#define W64_STORE_PAIR(orow_, grow_, inv_l_, store_, dt_, gp_, n0_, n1_, n2_, n3_, n4_, n5_, n6_, n7_)           \
  do {                                                                                                           \
    uint2 pk__[2];                                                                                               \
    f32x4 va__, vb__;                                                                                            \
    W64_AREAD4(va__, n0_, n1_, n2_, n3_);                                                                        \
    W64_AREAD4(vb__, n4_, n5_, n6_, n7_);                                                                        \
    _Pragma("unroll") for (int gg__ = 0; gg__ < 2; ++gg__) {                                                     \
      const int d0__ = (dt_) * 32 + 8 * (2 * (gp_) + gg__) + 4 * eh;                                             \
      f32x4 v__ = (gg__ ? vb__ : va__) * (inv_l_);                                                               \
      if (GATE) {                                                                                                \
        const f32x4 gt__ = Vec4<bf16_t>::load((grow_) + d0__);                                                   \
        _Pragma("unroll") for (int e__ = 0; e__ < 4; ++e__)                                                      \
          v__[e__] *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt__[e__] * -1.44269504088896340736f)); \
      }                                                                                                          \
      const bf16x4 b4__ = {(bf16_t)v__[0], (bf16_t)v__[1], (bf16_t)v__[2], (bf16_t)v__[3]};                      \
      pk__[gg__] = __builtin_bit_cast(uint2, b4__);                                                              \
    }                                                                                                            \
    /* lanes (r, 0) and (r, 1) hold the 4-feature groups 8 g + 0..3 and 8 g + 4..7 of a row: one v_permlane32_swap per dword   */ \
    /* leaves each lane with 8 consecutive features of the pair of groups -> one 16-byte store per lane (see k_attn_bf16)      */ \
    const auto sx__ = __builtin_amdgcn_permlane32_swap(pk__[0].x, pk__[1].x, false, false);                      \
    const auto sy__ = __builtin_amdgcn_permlane32_swap(pk__[0].y, pk__[1].y, false, false);                      \
    const uint4 o16__ = {sx__[0], sy__[0], sx__[1], sy__[1]};                                                    \
    if (store_) *reinterpret_cast<uint4*>((orow_) + (dt_) * 32 + 16 * (gp_) + 8 * eh) = o16__;                   \
  } while (0)
  // row indices from a fresh lane id: nothing lane-derived of the epilogue is live across the key loop
  const int elane = w64_fresh_lane();
  const int er = elane & 31, eh = elane >> 5;
  {
    const int qrow = q0 + er, qrc = qrow < S ? qrow : S - 1;
    const float l_tot = lA + __shfl_xor(lA, 32, 64);
    const float inv_l = __builtin_amdgcn_rcpf(l_tot);
    bf16_t* orow = out + (size_t)(s0 + qrc) * ldo + head * 64;
    const bf16_t* grow = gbase + (size_t)qrc * ld;
    const bool store = live && qrow < S;
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7);
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 1, 8, 9, 10, 11, 12, 13, 14, 15);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 0, 16, 17, 18, 19, 20, 21, 22, 23);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 1, 24, 25, 26, 27, 28, 29, 30, 31);
  }
  {
    const int qrow = q0 + 32 + er, qrc = qrow < S ? qrow : S - 1;
    const float l_tot = lB + __shfl_xor(lB, 32, 64);
    const float inv_l = __builtin_amdgcn_rcpf(l_tot);
    bf16_t* orow = out + (size_t)(s0 + qrc) * ldo + head * 64;
    const bf16_t* grow = gbase + (size_t)qrc * ld;
    const bool store = live && qrow < S;
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 0, 32, 33, 34, 35, 36, 37, 38, 39);
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 1, 40, 41, 42, 43, 44, 45, 46, 47);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 0, 48, 49, 50, 51, 52, 53, 54, 55);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 1, 56, 57, 58, 59, 60, 61, 62, 63);
  }
#ifdef ATTN64_STAMPS
  if (stamps && blockIdx.x % 37 == 0 && elane == 0) {      // [0..5] loop segments, [6] entry -> loop, [7] loop end -> stores issued
    unsigned long long st_end__;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_end__)::"memory");
    long long* dst = stamps + ((size_t)(blockIdx.x / 37) * 4 + wave) * 8;
    for (int i = 0; i < 6; ++i) dst[i] = (long long)st_acc__[i];
    dst[6] = (long long)(st_loop_start__ - st_entry__);
    dst[7] = (long long)(st_end__ - st_loop_end__);
  }
#endif
}

// items: device int32 [n_items, 8], see k_attn_w64.  Requirements (checked by the caller): bf16, head_dim 64, q pre-scaled.
int ttvk_attention64(const void* qkvg, int ld, void* out, int ldo, const int* cu_seqlens, const int* items, int n_items, int q_heads,
                     int kv_heads, int flags, hipStream_t s) {
  if (n_items == 0) return TTV_OK;
  TTV_CHECK_ARG(kv_heads > 0 && q_heads % kv_heads == 0 && q_heads <= 256, "attention64: q_heads %% kv_heads");
  const int d_model = q_heads * 64, gqa = kv_heads * 64, rep = q_heads / kv_heads;
  TTV_CHECK_ARG(ld >= 2 * d_model + 2 * gqa && ld % 8 == 0 && ldo % 8 == 0, "attention64: bad leading dims");
  TTV_CHECK_ARG((uintptr_t)qkvg % 16 == 0 && (uintptr_t)out % 16 == 0, "attention64: unaligned pointers");
  TTV_CHECK_ARG(flags & TTV_ATTN_QSCALED, "attention64: needs pre-scaled q (TTV_ATTN_QSCALED)");
  static const float defer_thr = getenv("TTV_ATTN_THR") ? (float)atof(getenv("TTV_ATTN_THR")) : 8.0f;
  TtvProfScope prof(TTV_KC_ATTENTION, s);
  if (flags & TTV_ATTN_GATE)
    hipLaunchKernelGGL((k_attn_w64<true>), dim3(n_items), dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, cu_seqlens, items, d_model, gqa,
                       rep, defer_thr, g_ttv_stamps);
  else
    hipLaunchKernelGGL((k_attn_w64<false>), dim3(n_items), dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, cu_seqlens, items, d_model, gqa,
                       rep, defer_thr, g_ttv_stamps);
  TTV_CHECK_LAUNCH("attention64");
  return TTV_OK;
}

// Variable-length, non-causal GQA attention, bf16, head_dim 64: the 64-query-rows-per-wave kernel (round 3).
// Replaces flash_attn_varlen_func at reference model/base/transformer.py:100 and the sigmoid gate at :103 for inference towers
// whose q columns arrive pre-scaled (ttv_layer_weights.qkv_q_prescaled); same arithmetic as k_attn_bf16<.., PRE = true> (ttv_attn.hip).
//
// Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure", adapted to head_dim 64):
//   * one workgroup = 4 waves = ONE wave per SIMD with the whole 512-entry register file; a wave owns 64 query rows (two 32-row
//     tiles A and B) of one q-head.  The four waves of a block work on the same (sequence, kv-head) - any q-head of that kv-head,
//     any 64-row slice - and share every K / V tile (work table built by the host, plan.attention_table64).
//   * every K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) read from LDS feeds TWO MFMAs (tile A and tile B):
//     half the LDS reads, LDS-DMA instructions, scalar bookkeeping and barriers per MFMA of the 32-rows-per-wave kernel.
//   * the two tiles run half a step apart, so that the matrix pipe and the vector unit of the SIMD always have independent work
//     from the SAME wave (tools/ubench/valu_rates.hip, corrected in round 3: an MFMA with 2 exp2 + 4..6 plain instructions in its
//     gap costs 41..46 cycles per SIMD whether they come from one wave or four - they do overlap, 32 + 40 would be 72):
//         iteration u (one 64-key tile), after its single barrier:
//           Ya:  PV_B(u-1)              beside  row maximum of S_A(u)        [+ LDS-DMA issue of K(u+3), V(u+2)]
//           --   rare: move tile A's softmax reference (wave-uniform branch; O_B's MFMAs may be in flight, O_A's are not)
//           Yb:  PV_B(u-1) rest, S_B(u) beside  exp2 / row sum / bf16 pack of S_A(u) -> P_A(u)     [+ V(u) fragment reads]
//           Xa:  PV_A(u)                beside  row maximum of S_B(u)        [+ K(u+1) fragment reads]
//           --   rare: move tile B's reference
//           Xb:  PV_A(u) rest, S_A(u+1) beside  exp2 / row sum / pack of S_B(u) -> P_B(u)
//     A reference shift of one tile never meets pending P.V MFMAs of the same tile (cdna_hip_programming.md T13 hazard): when
//     tile A's reference moves, PV_A(u-1) is complete and PV_A(u) not yet issued, and its P is computed afterwards.
//   * K / V tiles (64 keys) arrive by LDS-DMA (global_load_lds_dwordx4, XOR swizzle on the source side) into three-slot rings:
//     K three tiles ahead, V two; ONE raw s_barrier per tile behind a counted s_waitcnt vmcnt(4); fragments are read from LDS
//     half an iteration before the MFMAs that use them.
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

// work table: int32 [n_items, 8] = (sequence, kv-head, wave 0..3: q-head | (first query row / 64) << 8, or -1 = idle wave, 0, 0);
// sequence < 0 = padding entry of the XCD-interleaved order
template <bool GATE>
__global__ __launch_bounds__(256, 1) void k_attn_w64(const bf16_t* __restrict__ qkvg, int ld, bf16_t* __restrict__ out, int ldo,
                                                     const int* __restrict__ cu, const int* __restrict__ items, int d_model, int gqa, int rep,
                                                     float defer_thr, long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) uint4 kl[3][KB * 8];
  __shared__ __attribute__((aligned(16))) uint4 vl[3][KB * 8];
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
  const int s0 = cu[seq], S = cu[seq + 1] - s0;
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
#define W64_HALF(steady_, run_, base_, lds_, kt_, soff_, c_, d_, half_)                                          \
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
        const int lim__ = S - 1 - key0__;                                                                        \
        const int g__ = drow + 32 * (half_) < lim__ ? drow + 32 * (half_) : lim__;                               \
        W64_DMA((uint32_t)(g__ * ld + (c_)) * 2u, b__, dst__);                                                   \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)

  // lane-constant LDS addresses (layout of k_attn_bf16): the ring slot is a scalar byte offset, everything else an immediate
  // K fragment (A operand of S^T): key row 32 kb + r, 16-byte chunk (2 ks + h) ^ ((row >> 1) & 7)
  const int ksw = (r >> 1) & 7;
  const char* const kbase_lds = reinterpret_cast<const char*>(&kl[0][0]);
  const char* const vbase_lds = reinterpret_cast<const char*>(&vl[0][0]);
  const char* const ka0 = kbase_lds + r * 128 + (((0 * 2 + h) ^ ksw) << 4);
  const char* const ka1 = kbase_lds + r * 128 + (((1 * 2 + h) ^ ksw) << 4);
  const char* const ka2 = kbase_lds + r * 128 + (((2 * 2 + h) ^ ksw) << 4);
  const char* const ka3 = kbase_lds + r * 128 + (((3 * 2 + h) ^ ksw) << 4);
  // V^T fragment via ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its block
  const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
  const int vsw = (tq >> 1) & 1;
  const int vlane = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
  const char* const va0 = vbase_lds + vlane + (vsw ? 64 : 0);   // head dims 0..31
  const char* const va1 = vbase_lds + vlane + (vsw ? 0 : 64);   // head dims 32..63
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
  f32x16 sA0, sA1, sB0 = zero16, sB1 = zero16;                     // score tiles: tile x 32-key half
  f32x16 negmA = zero16, negmB = zero16;                           // start vectors of the score accumulators: -m per query (lane)
  float mA = 0.f, mB = 0.f, lA = 0.f, lB = 0.f;
  bf16x8 pA0, pA1, pA2, pA3;                                        // P as B fragments, k-step (kb, sp) -> index 2 kb + sp
  bf16x8 pB0 = zero8, pB1 = zero8, pB2 = zero8, pB3 = zero8;
  bf16x8 kf00, kf01, kf02, kf03, kf10, kf11, kf12, kf13;            // K fragments [kb][ks] of the tile S is computed from next
  bf16x8 vf000 = zero8, vf001 = zero8, vf010 = zero8, vf011 = zero8, vf100 = zero8, vf101 = zero8, vf110 = zero8, vf111 = zero8;   // V^T [dt][kb][sp]

  // Every MFMA is an asm statement: the compiler, left to choose, accumulates the score tiles in AGPRs too (one wave per SIMD makes
  // the accumulation file available) and then moves every score through v_accvgpr_read before the softmax - 576 copies per
  // iteration in the first build.  Here the O^T accumulators are pinned to AGPRs ("+a": touched only by MFMAs, the rare reference
  // shift and the epilogue), the score tiles to VGPRs ("v").  An asm statement is opaque to the scheduler, so the interleave is
  // written out: a SLOT is one MFMA followed by a chunk of vector work, fenced with sched_barrier(0).
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
  // K fragments [kb][ks] of ring slot soff_, two reads per call (i_ = 0..3: literal)
#define W64_LOADK2(soff_, i_)                                                                                    \
  do {                                                                                                           \
    if (W64_KO_LDS) break;                                                                                       \
    if ((i_) == 0) { kf00 = *reinterpret_cast<const bf16x8*>(ka0 + (soff_)); kf10 = *reinterpret_cast<const bf16x8*>(ka0 + (soff_) + 4096); } \
    if ((i_) == 1) { kf01 = *reinterpret_cast<const bf16x8*>(ka1 + (soff_)); kf11 = *reinterpret_cast<const bf16x8*>(ka1 + (soff_) + 4096); } \
    if ((i_) == 2) { kf02 = *reinterpret_cast<const bf16x8*>(ka2 + (soff_)); kf12 = *reinterpret_cast<const bf16x8*>(ka2 + (soff_) + 4096); } \
    if ((i_) == 3) { kf03 = *reinterpret_cast<const bf16x8*>(ka3 + (soff_)); kf13 = *reinterpret_cast<const bf16x8*>(ka3 + (soff_) + 4096); } \
  } while (0)
#define W64_VFRAG(va_, OFF_)                                                                                     \
  ({                                                                                                             \
    const bf16x4 lo__ = lds64_read_tr16((va_) + (OFF_)), hi__ = lds64_read_tr16((va_) + (OFF_) + 1024);          \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                            \
  })
  // V^T fragments [dt][kb][sp] of ring slot soff_, four reads (two fragments) per call (i_ = 0..3: literal)
#define W64_LOADV4(soff_, i_)                                                                                    \
  do {                                                                                                           \
    if (W64_KO_LDS) break;                                                                                       \
    const char* v0__ = va0 + (soff_);                                                                            \
    const char* v1__ = va1 + (soff_);                                                                            \
    if ((i_) == 0) { vf000 = W64_VFRAG(v0__, 0);    vf100 = W64_VFRAG(v1__, 0); }                                \
    if ((i_) == 1) { vf001 = W64_VFRAG(v0__, 2048); vf101 = W64_VFRAG(v1__, 2048); }                             \
    if ((i_) == 2) { vf010 = W64_VFRAG(v0__, 4096); vf110 = W64_VFRAG(v1__, 4096); }                             \
    if ((i_) == 3) { vf011 = W64_VFRAG(v0__, 6144); vf111 = W64_VFRAG(v1__, 6144); }                             \
  } while (0)
  // keys past the end of the sequence (last tile of a sequence whose length is not a multiple of 64)
#define W64_MASK(s0_, s1_, kt_)                                                                                  \
  do {                                                                                                           \
    if ((kt_) == nkt - 1 && nkt * KB > S) {                                                                      \
      int lim__ = S - (kt_) * KB - 4 * h;             /* opaque: keeps 32 precomputed lane masks out of the loop's scalar registers */ \
      asm volatile("" : "+v"(lim__));                                                                            \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) {                                                     \
        const int key__ = (e__ & 3) + 8 * (e__ >> 2);                                                            \
        if (key__ >= lim__) s0_[e__] = -INFINITY;                                                                \
        if (key__ + 32 >= lim__) s1_[e__] = -INFINITY;                                                           \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
  // row maximum of the 64 scores of a query: chain c_ (literal 0..3) over 8 registers, then the combine over both lane halves
#define W64_MAXCHAIN(dst_, s_, b_)                                                                               \
  do {                                                                                                           \
    if (W64_KO_MAX) { dst_ = s_[(b_)]; break; }                                                                  \
    float a__ = fmaxf(fmaxf(s_[(b_) + 0], s_[(b_) + 1]), s_[(b_) + 2]);                                          \
    a__ = fmaxf(fmaxf(a__, s_[(b_) + 3]), s_[(b_) + 4]);                                                         \
    a__ = fmaxf(fmaxf(a__, s_[(b_) + 5]), s_[(b_) + 6]);                                                         \
    dst_ = fmaxf(a__, s_[(b_) + 7]);                                                                             \
  } while (0)
#define W64_MAXFIN(c0_, c1_, c2_, c3_)                                                                           \
  ({                                                                                                             \
    const float m2__ = fmaxf(fmaxf(fmaxf(c0_, c1_), c2_), c3_);                                                  \
    const auto sw__ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2__), __builtin_bit_cast(unsigned, m2__), false, false); \
    fmaxf(__builtin_bit_cast(float, sw__[0]), __builtin_bit_cast(float, sw__[1]));                               \
  })
  // the tile's scores are relative to its running reference (they started from -m).  The reference moves only when some score of
  // the tile exceeds it by more than defer_thr (first tile: always, the reference is the placeholder 0): scores, row sum, O and
  // the start vector are shifted (see k_attn_bf16)
#define W64_SHIFT(first_, mx_, s0_, s1_, OSCALE_, l_, m_, negm_)                                                 \
  do {                                                                                                           \
    if (W64_KO_MAX) break;                                                                                       \
    if ((first_) || __builtin_amdgcn_ballot_w64((mx_) > defer_thr) != 0ull) {                                    \
      const float d__ = (first_) ? (mx_) : fmaxf((mx_), 0.f);                                                    \
      if (!(first_)) {                                            /* nothing accumulated yet on the first tile */ \
        const float alpha__ = __builtin_amdgcn_exp2f(-d__);                                                      \
        l_ *= alpha__;                                                                                           \
        OSCALE_(alpha__);                                                                                        \
      }                                                                                                          \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) {                                                     \
        s0_[e__] -= d__;                                                                                         \
        s1_[e__] -= d__;                                                                                         \
      }                                                                                                          \
      m_ += d__;                                                                                                 \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) negm_[e__] = -m_;                                     \
      asm volatile("s_nop 1" ::: "memory");                                                                      \
    }                                                                                                            \
  } while (0)
  // softmax of one tile in eight chunks of four scores (chunk c_: registers 4 (c_ & 3) .. + 3 of s0_ (c_ < 4) or s1_): E = exp2 in
  // place, A = add to the four partial row sums, C = pack to bf16 into the P fragment 2 kb + sp = c_ >> 1
#define W64_SREG(s0_, s1_, c_, j_) ((c_) < 4 ? s0_[4 * (c_) + (j_)] : s1_[4 * ((c_) - 4) + (j_)])
#define W64_E(s0_, s1_, c_)                                                                                      \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) {                                                        \
      if (W64_KO_EXP) continue;                                                                                  \
      if ((c_) < 4) s0_[4 * ((c_) & 3) + j__] = __builtin_amdgcn_exp2f(s0_[4 * ((c_) & 3) + j__]);               \
      else s1_[4 * ((c_) & 3) + j__] = __builtin_amdgcn_exp2f(s1_[4 * ((c_) & 3) + j__]);                        \
    }                                                                                                            \
    /* pin: the results exist HERE (pure arithmetic is otherwise sunk into the block of its first use, past every fence) */ \
    if ((c_) < 4) asm volatile("" : "+v"(s0_)); else asm volatile("" : "+v"(s1_));                               \
  } while (0)
#define W64_A(s0_, s1_, c_)                                                                                      \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) {                                                        \
      if ((c_) == 0) ps__[j__] = W64_SREG(s0_, s1_, c_, j__);                                                    \
      else ps__[j__] += W64_SREG(s0_, s1_, c_, j__);                                                             \
    }                                                                                                            \
    asm volatile("" : "+v"(ps__[0]), "+v"(ps__[1]), "+v"(ps__[2]), "+v"(ps__[3]));                               \
  } while (0)
#define W64_C(s0_, s1_, c_, pf_)                                                                                 \
  do {                                                                                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) pf_[4 * ((c_) & 1) + j__] = (bf16_t)W64_SREG(s0_, s1_, c_, j__);   \
    asm volatile("" : "+v"(pf_));                                                                                \
  } while (0)
  // ---- prologue: K(0) | V(0), K(1) | K(2), V(1) in flight (10 DMA instructions per wave); S_A(0) ----
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
  W64_LOADK2(0, 0); W64_LOADK2(0, 1); W64_LOADK2(0, 2); W64_LOADK2(0, 3);
  W64_FENCE();
  W64_MFMA_S0(sA0, kf00, 64, 67, negmA);
  W64_MFMA_S0(sA1, kf10, 64, 67, negmA);
  W64_MFMA_S(sA0, kf01, 68, 71);
  W64_MFMA_S(sA1, kf11, 68, 71);
  W64_MFMA_S(sA0, kf02, 72, 75);
  W64_MFMA_S(sA1, kf12, 72, 75);
  W64_MFMA_S(sA0, kf03, 76, 79);
  W64_MFMA_S(sA1, kf13, 76, 79);
  W64_STAMP_DECL;
  W64_STAMP_START();
#ifdef ATTN64_STAMPS
  const unsigned long long st_loop_start__ = st_prev__;
#endif

  int o0 = 0, o1 = KB * 128, o2 = 2 * KB * 128;          // ring slot byte offsets of tiles u, u+1, u+2 (scalar registers)
  for (int u = 0; u < nkt; ++u) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // K(u+1), V(u) landed (own share); K(u+2), V(u+1) may fly
    W64_STAMP(0);
    __builtin_amdgcn_s_barrier();                        // ... for every wave; all are done reading K(u) and V(u-1) from LDS
    W64_STAMP(1);
    W64_MASK(sA0, sA1, u);
    W64_FENCE();

    // K(u+3) -> slot of K(u), V(u+2) -> slot of V(u-1): one DMA instruction per region
    const bool steady = u + 3 < nkt - 1;

    // ---- Ya: PV_B(u-1) | row maximum of S_A(u)
    float c0, c1, c2, c3;
    W64_MFMA_OB0(vf000, pB0);
    W64_HALF(steady, kdma, kbase, kl_lds, u + 3, o0, kc, dK, 0);
    W64_FENCE();
    W64_MFMA_OB1(vf100, pB0);
    W64_MAXCHAIN(c0, sA0, 0);
    W64_FENCE();
    W64_MFMA_OB0(vf001, pB1);
    W64_MAXCHAIN(c1, sA0, 8);
    W64_FENCE();
    W64_MFMA_OB1(vf101, pB1);
    W64_MAXCHAIN(c2, sA1, 0);
    W64_FENCE();
    W64_MFMA_OB0(vf010, pB2);
    W64_MAXCHAIN(c3, sA1, 8);
    W64_FENCE();
    W64_MFMA_OB1(vf110, pB2);
    const float mxA = W64_MAXFIN(c0, c1, c2, c3);
    W64_FENCE();
    W64_MFMA_OB0(vf011, pB3);
    W64_MFMA_OB1(vf111, pB3);
    W64_SHIFT(u == 0, mxA, sA0, sA1, W64_OSCALE_A, lA, mA, negmA);
    W64_FENCE();
    W64_STAMP(2);

    // ---- Yb: V(u) fragments; S_B(u) | exp2 / row sum / pack of S_A(u)
    {
      float ps__[4];
      W64_MFMA_S0(sB0, kf00, 80, 83, negmB);
      W64_LOADV4(o0, 0);
      W64_E(sA0, sA1, 0);
      W64_FENCE();
      W64_MFMA_S0(sB1, kf10, 80, 83, negmB);
      W64_LOADV4(o0, 1);
      W64_E(sA0, sA1, 1); W64_A(sA0, sA1, 0);
      W64_FENCE();
      W64_MFMA_S(sB0, kf01, 84, 87);
      W64_LOADV4(o0, 2);
      W64_E(sA0, sA1, 2); W64_A(sA0, sA1, 1); W64_C(sA0, sA1, 0, pA0);
      W64_FENCE();
      W64_MFMA_S(sB1, kf11, 84, 87);
      W64_LOADV4(o0, 3);
      W64_E(sA0, sA1, 3); W64_A(sA0, sA1, 2); W64_C(sA0, sA1, 1, pA0);
      W64_FENCE();
      W64_MFMA_S(sB0, kf02, 88, 91);
      W64_HALF(steady, kdma, kbase, kl_lds, u + 3, o0, kc, dK, 1);
      W64_E(sA0, sA1, 4); W64_A(sA0, sA1, 3); W64_C(sA0, sA1, 2, pA1);
      W64_FENCE();
      W64_MFMA_S(sB1, kf12, 88, 91);
      W64_E(sA0, sA1, 5); W64_A(sA0, sA1, 4); W64_C(sA0, sA1, 3, pA1);
      W64_FENCE();
      W64_MFMA_S(sB0, kf03, 92, 95);
      W64_E(sA0, sA1, 6); W64_A(sA0, sA1, 5); W64_C(sA0, sA1, 4, pA2);
      W64_FENCE();
      W64_MFMA_S(sB1, kf13, 92, 95);
      W64_E(sA0, sA1, 7); W64_A(sA0, sA1, 6); W64_C(sA0, sA1, 5, pA2);
      W64_FENCE();
      W64_A(sA0, sA1, 7); W64_C(sA0, sA1, 6, pA3);
      W64_C(sA0, sA1, 7, pA3);
      lA += (ps__[0] + ps__[1]) + (ps__[2] + ps__[3]);
      asm volatile("" : "+v"(lA));
      W64_FENCE();
    }
    W64_STAMP(3);
    W64_MASK(sB0, sB1, u);
    W64_FENCE();

    // ---- Xa: PV_A(u); K(u+1) fragments | row maximum of S_B(u)
    W64_MFMA_OA0(vf000, pA0);
    W64_HALF(steady, vdma, vbase, vl_lds, u + 2, o2, vc, dV, 0);
    W64_FENCE();
    W64_MFMA_OA1(vf100, pA0);
    W64_LOADK2(o1, 0);
    W64_MAXCHAIN(c0, sB0, 0);
    W64_FENCE();
    W64_MFMA_OA0(vf001, pA1);
    W64_LOADK2(o1, 1);
    W64_MAXCHAIN(c1, sB0, 8);
    W64_FENCE();
    W64_MFMA_OA1(vf101, pA1);
    W64_LOADK2(o1, 2);
    W64_MAXCHAIN(c2, sB1, 0);
    W64_FENCE();
    W64_MFMA_OA0(vf010, pA2);
    W64_LOADK2(o1, 3);
    W64_MAXCHAIN(c3, sB1, 8);
    W64_FENCE();
    W64_MFMA_OA1(vf110, pA2);
    const float mxB = W64_MAXFIN(c0, c1, c2, c3);
    W64_FENCE();
    W64_MFMA_OA0(vf011, pA3);
    W64_MFMA_OA1(vf111, pA3);
    W64_SHIFT(u == 0, mxB, sB0, sB1, W64_OSCALE_B, lB, mB, negmB);
    W64_FENCE();
    W64_STAMP(4);

    // ---- Xb: S_A(u+1) | exp2 / row sum / pack of S_B(u)
    {
      float ps__[4];
      W64_MFMA_S0(sA0, kf00, 64, 67, negmA);      // past the last tile: scores of the re-fetched tile, unused
      W64_E(sB0, sB1, 0);
      W64_FENCE();
      W64_MFMA_S0(sA1, kf10, 64, 67, negmA);
      W64_E(sB0, sB1, 1); W64_A(sB0, sB1, 0);
      W64_FENCE();
      W64_MFMA_S(sA0, kf01, 68, 71);
      W64_E(sB0, sB1, 2); W64_A(sB0, sB1, 1); W64_C(sB0, sB1, 0, pB0);
      W64_FENCE();
      W64_MFMA_S(sA1, kf11, 68, 71);
      W64_E(sB0, sB1, 3); W64_A(sB0, sB1, 2); W64_C(sB0, sB1, 1, pB0);
      W64_FENCE();
      W64_MFMA_S(sA0, kf02, 72, 75);
      W64_HALF(steady, vdma, vbase, vl_lds, u + 2, o2, vc, dV, 1);
      W64_E(sB0, sB1, 4); W64_A(sB0, sB1, 3); W64_C(sB0, sB1, 2, pB1);
      W64_FENCE();
      W64_MFMA_S(sA1, kf12, 72, 75);
      W64_E(sB0, sB1, 5); W64_A(sB0, sB1, 4); W64_C(sB0, sB1, 3, pB1);
      W64_FENCE();
      W64_MFMA_S(sA0, kf03, 76, 79);
      W64_E(sB0, sB1, 6); W64_A(sB0, sB1, 5); W64_C(sB0, sB1, 4, pB2);
      W64_FENCE();
      W64_MFMA_S(sA1, kf13, 76, 79);
      W64_E(sB0, sB1, 7); W64_A(sB0, sB1, 6); W64_C(sB0, sB1, 5, pB2);
      W64_FENCE();
      W64_A(sB0, sB1, 7); W64_C(sB0, sB1, 6, pB3);
      W64_C(sB0, sB1, 7, pB3);
      lB += (ps__[0] + ps__[1]) + (ps__[2] + ps__[3]);
      asm volatile("" : "+v"(lB));
      W64_FENCE();
    }
    kdma += (size_t)KB * ld;
    vdma += (size_t)KB * ld;
    W64_STAMP(5);

    const int o = o0;
    o0 = o1;
    o1 = o2;
    o2 = o;
  }
  // ---- drain: PV_B(nkt-1) (its V fragments were read in the last Yb) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the block's LDS is released
  W64_MFMA_OB0(vf000, pB0);
  W64_MFMA_OB1(vf100, pB0);
  W64_MFMA_OB0(vf001, pB1);
  W64_MFMA_OB1(vf101, pB1);
  W64_MFMA_OB0(vf010, pB2);
  W64_MFMA_OB1(vf110, pB2);
  W64_MFMA_OB0(vf011, pB3);
  W64_MFMA_OB1(vf111, pB3);
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
      const int d0__ = (dt_) * 32 + 8 * (2 * (gp_) + gg__) + 4 * h;                                              \
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
    if (store_) *reinterpret_cast<uint4*>((orow_) + (dt_) * 32 + 16 * (gp_) + 8 * h) = o16__;                    \
  } while (0)
  {
    const float l_tot = lA + __shfl_xor(lA, 32, 64);
    const float inv_l = __builtin_amdgcn_rcpf(l_tot);
    bf16_t* orow = out + (size_t)(s0 + qrcA) * ldo + head * 64;
    const bf16_t* grow = gbase + (size_t)qrcA * ld;
    const bool store = live && qrowA < S;
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7);
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 1, 8, 9, 10, 11, 12, 13, 14, 15);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 0, 16, 17, 18, 19, 20, 21, 22, 23);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 1, 24, 25, 26, 27, 28, 29, 30, 31);
  }
  {
    const float l_tot = lB + __shfl_xor(lB, 32, 64);
    const float inv_l = __builtin_amdgcn_rcpf(l_tot);
    bf16_t* orow = out + (size_t)(s0 + qrcB) * ldo + head * 64;
    const bf16_t* grow = gbase + (size_t)qrcB * ld;
    const bool store = live && qrowB < S;
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 0, 32, 33, 34, 35, 36, 37, 38, 39);
    W64_STORE_PAIR(orow, grow, inv_l, store, 0, 1, 40, 41, 42, 43, 44, 45, 46, 47);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 0, 48, 49, 50, 51, 52, 53, 54, 55);
    W64_STORE_PAIR(orow, grow, inv_l, store, 1, 1, 56, 57, 58, 59, 60, 61, 62, 63);
  }
#ifdef ATTN64_STAMPS
  if (stamps && blockIdx.x % 37 == 0 && lane == 0) {      // [0..5] loop segments, [6] entry -> loop, [7] loop end -> stores issued
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

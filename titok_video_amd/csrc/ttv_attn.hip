// Variable-length, non-causal GQA attention with the sigmoid output gate fused
// (replaces flash_attn_varlen_func at reference model/base/transformer.py:100 and the gate at :103).
//
// bf16 kernel (head_dim 64).  One workgroup = 4 waves = 128 query rows of one (sequence, q-head); each wave owns
// 32 queries.  K/V tiles of 64 keys are staged by LDS-DMA into double-buffered, XOR-swizzled LDS (157 VGPRs with pre-scaled q and the gate, no AGPRs: 3 blocks per CU).
// "Half items" (work-table mode 1): 64 query rows, wave pairs split the key range and merge their (O, m, l) at the end.
//   S^T = K Q^T   : mfma_f32_32x32x16_bf16 with the KEY on the MFMA row and the QUERY on the lane (col = lane&31),
//                   so a lane holds 32 scores of ONE query: row max / row sum are in-lane plus one xor-32 exchange.
//   O^T = V^T P^T : the S^T accumulator registers, packed to bf16, ARE the B operand (k order
//                   16s + 8(j>>2) + 4h + (j&3)); the matching V^T A operand is fetched with ds_read_b64_tr_b16
//                   (hardware transpose read) straight from the row-major [key][d] LDS tile.
// Online softmax in fp32 (exp2 with the scale folded into one FMA); output is normalised, multiplied by
// sigmoid(gate) and stored as 4 consecutive head dims per lane.
//
// fp32 kernel: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), same block decomposition; see k_attn_f32.
#include <stdlib.h>

#include "ttv_common.h"
#include "ttv_kernels.h"

#define QB 128
#define KB 64
// online softmax: the running reference of a query row moves only when a score exceeds it by more than this (log2 units)
#define ATTN_DEFER_THR 8.0f
// pre-scaled-q kernel: the row maximum is taken only when a tile's row sums say the running reference is stale (see LAZY in k_attn_bf16)
#ifndef ATTN_LAZY
#define ATTN_LAZY 1
#endif
#ifndef ATTN_DMA_LATE
#define ATTN_DMA_LATE 1
#endif
#ifndef ATTN_PRIO
#define ATTN_PRIO 0     // measured neutral either way (profiles/r04_attn_variants.txt); 1: the softmax phase of a wave runs at raised issue priority (s_setprio), 2: the MFMA phases do, 0: neither
#endif
#define ATTN_LAZY_LOG2 30
#define ATTN_LAZY_BIG 1073741824.0f
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements is returned column-major
// (lane i of the group gets column i of the 4 rows).  EXEC must be all ones; addresses 8-byte aligned.
__device__ __forceinline__ bf16x4 lds_read_tr16(const char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds_ptr));
}

// Diagnostic build only (-DATTN_STAMPS, tools/attn_stamps.py): s_memtime stamps at the segment boundaries of the key loop of full
// items, summed per wave over the loop and written by every 37th block to stamps[(block / 37) * 4 + wave][8].  The product
// library is built without it (no stamp executes, no fence constrains the scheduler).
#ifdef ATTN_STAMPS
#define STAMP_DECL unsigned long long st_prev__ = 0, st_acc__[6] = {0, 0, 0, 0, 0, 0}
#define STAMP_START()                                                                                  \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev__)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
#define STAMP(seg_)                                                                                    \
  do {                                                                                                 \
    unsigned long long t__;                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    st_acc__[seg_] += t__ - st_prev__;                                                                 \
    st_prev__ = t__;                                                                                   \
  } while (0)
#else
#define STAMP_DECL
#define STAMP_START()
#define STAMP(seg_)
#endif

// Diagnostic build -DATTN_TIMELINE (tools/attn_timeline.py): wave 0 of every block writes, per table entry, the constant 100 MHz clock
// (s_memrealtime) and the shader clock (s_memtime) at entry start / loop start / loop end / entry end plus the HW_ID and XCC_ID registers
// to stamps[entry][8]: when and where every block ran, and the shader clock the part held meanwhile.
#ifdef ATTN_TIMELINE
#define TL_READ(real_, core_)                                                                          \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real_), "=s"(core_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
#define TL_DECL unsigned long long tl_r__[4] = {0, 0, 0, 0}, tl_c__[4] = {0, 0, 0, 0}
#define TL_MARK(i_) TL_READ(tl_r__[i_], tl_c__[i_])
#else
#define TL_DECL
#define TL_MARK(i_)
#endif

// NE = table entries per block.  NE == 2 ("paired" tables, plan.attention_table): a block of 8 waves takes entries 2j and 2j+1 of
// its XCD list - the same query rows of the two q-heads that share a kv-head - so every K / V tile is staged ONCE for both heads
// (half the tile traffic through L2 and LDS per score).  Waves 0-3 work on the first entry, 4-7 on the second, exactly as the
// four waves of an NE == 1 block; only the staging is split eight ways.
// PRE: the q columns arrive multiplied by scale * log2(e) (folded into the q rows of the projection weight by the host, one rounding,
// ttv_layer_weights.qkv_q_prescaled): S is then already the exponent.  The running maximum is carried INSIDE the MFMA accumulator -
// the score tiles start from -m instead of 0 - so a score needs no multiply-subtract before its exp2 (4 instead of 5 issue slots);
// when a tile raises the maximum (rare after the first tiles) the tile's scores, the running state and the start vector are shifted.
// TAPE: the training forward - additionally writes the log-sum-exp per (row, head) and, with the gate, the ungated output.
//
// What bounds it (round-3 measurements; DESIGN.md section 4 has the table):
//   * the shader clock the part holds under this kernel is 1.8-1.9 GHz, not the 2.4 GHz the 2.5 PFLOP/s peak is quoted at
//     (s_memtime against s_memrealtime inside the kernel, -DATTN_TIMELINE, profiles/r03_attn_timeline.txt);
//   * at three waves per SIMD a wave needs ~2 650 cycles per 64-key tile (profiles/r03_attn_stamps.txt), i.e. a SIMD retires a
//     32-query x 64-key unit every ~885 cycles against the 512 cycles of its 16 MFMAs: the unit also issues ~90 plain vector
//     instructions (4 cycles each), 32 v_exp_f32 (8 each) and the MFMAs hold the issue port for 8 cycles each - 128 + 256 + 360
//     = ~750 cycles of issue on a SIMD whose matrix and vector pipes overlap only partly (tools/ubench/valu_rates.hip).  Row sums
//     on the matrix pipe (-DATTN_MSUM) trade 32 adds for 4 MFMAs and are 5 % SLOWER (65.0 vs 61.7 us);
//   * a SIMD's throughput barely depends on how many waves it holds - 9.2 us per wave and entry at three waves, 10.4 at two, 20.2
//     for a lone wave - and the three co-resident blocks of a CU finish staggered (22 / 32 / 42 us: vector issue is arbitrated by
//     age), so a launch is NOT a sequence of rounds: its time grows linearly with the table (8 us + 45.5 ns per entry from 252 to
//     2 304 entries, profiles/r03_attn_staircase.txt) and the last blocks to start run on nearly empty SIMDs (the tail);
//   * a persistent walk of the table by 768 resident blocks (tried, tests green) is no faster, 63.4 vs 61.5 us: a fresh block's
//     prologue is 0.8 us and its epilogue + store acknowledgement 2.5 us of 29 us, and other blocks issue meanwhile.
// Round 4 (tools/ubench/valu_rates.hip PINGPONG=1, tools/ubench/lds_rates.hip; profiles/r04_pingpong.txt, r04_lds_rates.txt,
// r04_attn_variants.txt):
//   * matrix and vector work of DIFFERENT waves of a SIMD do not overlap: two waves of one block run in forced anti-phase (one
//     issues its 16 MFMAs while the other issues its vector phase, block barriers between phases) need 1 155 cycles per phase
//     for 514 cycles of MFMAs beside ~510 of vector work - the sum -, unsynchronised co-resident waves 780-900 per unit.  Only
//     vector instructions placed behind an MFMA in the SAME wave run under it (43-46 cycles per MFMA + 2 exp2 + 6 plain).  So
//     per unit a SIMD of this kernel pays MFMA issue (512) + vector issue serially: 885 cycles measured, and a wave's softmax is
//     what the loop can still shed;
//   * the loop's LDS fragment traffic (8 ds_read_b128 + 16 ds_read_b64_tr_b16 per tile and wave) takes ~310 cycles per unit and
//     SIMD at the 210-240 bytes / cycle / CU the part delivers: a third of the loop, not its bound;
//   * hence LAZY below (no row maximum per tile; 113 -> 76 vector instructions per tile with the packed row sums): -5 % at the
//     benchmark's score spread, -12 % at a spread of 6 (where the deferred reference still moved).  Four blocks per CU on 128
//     registers (+3 %), the next tile's DMA behind the score MFMAs instead of behind the barrier, s_setprio by phase in either
//     direction: neutral to worse, same box.
// Everything off the wave's chain of dependent phases that could be moved has been:
//   * the running reference of the softmax moves only when a score exceeds it by more than `defer_thr` (see below);
//   * the DMA source of a tile is a scalar base (advanced per tile on the scalar unit) plus lane-constant offsets: no per-tile
//     address arithmetic on the vector unit except in a sequence's last, partial tile;
//   * the epilogue uses v_rcp_f32 for 1/l and the sigmoid (a full-precision division is ~10 instructions per element and the
//     epilogue was a quarter of the wave's vector instructions) and stores 16 bytes per lane (lanes l, l+32 exchange 8-byte
//     groups with v_permlane32_swap).
// MXO (round 4, config #5): the output leaves ONLY as the block-scaled e4m3 operand of out_proj - `out_raw` then points at the fp8 image
// [L, d_model], `lse_out` at its E8M0 scales (k_quant_mx_fp8's layout; lse_out[0 .. ] reinterpreted, `ldo` = bytes of scales per row) -:
// a 32-element block of a row is one d-half of one head, held by the two lanes (r, 0), (r, 1) of a query, so its maximum is one
// lane exchange away; the bf16 output and the quantisation pass over it disappear.  The loop is the same code as the plain instantiation.
template <bool GATE, int NE, bool PRE, bool TAPE, bool MXO = false>
__global__ __launch_bounds__(256 * NE, NE == 2 ? 4 : 3) void k_attn_bf16(const bf16_t* __restrict__ qkvg, int ld, bf16_t* __restrict__ out, int ldo,
                                                           const int* __restrict__ cu, const int* __restrict__ qblocks, int n_entries,
                                                           int d_model, int gqa, int rep, float c_exp /* scale*log2(e) */,
                                                           float* __restrict__ lse_out, bf16_t* __restrict__ out_raw, float defer_thr, long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) uint4 kl[2][KB * 8];
  __shared__ __attribute__((aligned(16))) uint4 vl[2][KB * 8];
  __shared__ float xm_s[NE * 2 * 2 * 64];       // half items: (m, l) hand-over of the second wave pair of each entry

  const int tid = threadIdx.x, lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave in the block (0 .. 4 NE - 1), scalar: staging share
  const int ent = NE == 2 ? (w8 >> 2) : 0;       // which of the block's entries this wave works on
  const int wave = w8 & 3;                       // wave within the entry
  const int r = lane & 31, h = lane >> 5;
  {
  // work table entry: (sequence, first query row, q-head, mode); sequence < 0 = padding entry of the XCD-interleaved order.
  // Paired: entries (2j, 2j+1) of list x = blockIdx % 8 sit at flat rows (2j) * 8 + x and (2j + 1) * 8 + x.
  int tix = blockIdx.x;
  bool live = true;
  TL_DECL;
  TL_MARK(0);
  if (NE == 2) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int ta = (2 * j) * 8 + x, tb = ta + 8;
    const int sa = ta < n_entries ? qblocks[4 * ta] : -1, sb = tb < n_entries ? qblocks[4 * tb] : -1;
    if (sa < 0 && sb < 0) return;
    live = (ent ? sb : sa) >= 0;
    // an odd list length leaves one entry without a partner: its second wave quad shadows it (same staging share and barriers)
    // and stores nothing
    tix = live ? (ent ? tb : ta) : (ent ? ta : tb);
  }
  const int seq = qblocks[4 * tix], q0 = qblocks[4 * tix + 1], head = qblocks[4 * tix + 2];
  // mode 1 = "half item": 64 query rows; waves 0,1 take the first half of the key range, waves 2,3 the second half (32-key
  // tiles, one per wave and step) and the two partial (O, m, l) states are merged through LDS at the end.  A half item takes
  // about 0.6 of the time of a full one; the host puts them at the end of the table where they fill the tail of the grid
  // (plan.attention_table).
  const int mode = qblocks[4 * tix + 3];
  if (seq < 0) return;
  const int s0 = cu[seq], S = cu[seq + 1] - s0;
  const int kvh = head / rep;
  const bf16_t* qbase = qkvg + (size_t)s0 * ld + head * 64;
  const bf16_t* gbase = qkvg + (size_t)s0 * ld + d_model + head * 64;
  const bf16_t* kbase = qkvg + (size_t)s0 * ld + 2 * d_model + kvh * 64;
  const bf16_t* vbase = kbase + gqa;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query r][16*ks + 8h + 0..7]
  const int qrow = q0 + (mode ? (wave & 1) : wave) * 32 + r;
  const int qrc = qrow < S ? qrow : S - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrc * ld + ks * 16 + h * 8);

  // K / V tiles are staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, nothing to wait for before
  // the barrier except the DMA itself).  Instruction i of wave w covers tile rows 8 (2w + i) .. + 7: lane >> 3 picks the row,
  // lane & 7 the 16-byte LDS chunk; the DMA writes lanes linearly, so the XOR swizzles are applied on the global side (the lane
  // fetches the chunk that belongs at its LDS position): K chunk c holds global chunk c ^ ((row >> 1) & 7), V chunk c holds
  // c ^ (((row >> 1) & 1) << 2).  Rows past the sequence end re-fetch its last row (masked in the scores).
  const int wave_s = __builtin_amdgcn_readfirstlane(w8);       // staging share: wave w8 of 4 NE
  const uint32_t kl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&kl[0][0];
  const uint32_t vl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&vl[0][0];
  constexpr int DPW = 2 / NE;                                   // DMA instructions per wave, tile and operand
  const int drow0 = (wave_s * DPW) * 8 + (lane >> 3), drow1 = drow0 + 8;
  const int kc0 = ((lane & 7) ^ ((drow0 >> 1) & 7)) * 8, kc1 = ((lane & 7) ^ ((drow1 >> 1) & 7)) * 8;
  const int vc0 = ((lane & 7) ^ (((drow0 >> 1) & 1) << 2)) * 8, vc1 = ((lane & 7) ^ (((drow1 >> 1) & 1) << 2)) * 8;
#define DMA16(voff_, base_, dst_)                                                                                \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
  // tile rows 0..31 <- keys base0_ + row, rows 32..63 <- keys base1_ + (row - 32)   (waves 0,1 stage the first, 2,3 the second half)
#define DMA2(base0_, base1_, buf_)                                                                               \
  do {                                                                                                           \
    const int rb__ = wave_s < 2 * NE ? (base0_) : (base1_) - 32;                                                 \
    int g0__ = rb__ + drow0, g1__ = rb__ + drow1;                                                                \
    g0__ = g0__ < S ? g0__ : S - 1;                                                                              \
    g1__ = g1__ < S ? g1__ : S - 1;                                                                              \
    const uint32_t dk__ = kl_lds + (buf_) * (KB * 128) + wave_s * (1024 * DPW);                                  \
    const uint32_t dv__ = vl_lds + (buf_) * (KB * 128) + wave_s * (1024 * DPW);                                  \
    DMA16((uint32_t)(g0__ * ld + kc0) * 2u, kbase, dk__);                                                        \
    if (DPW == 2) DMA16((uint32_t)(g1__ * ld + kc1) * 2u, kbase, dk__ + 1024);                                   \
    DMA16((uint32_t)(g0__ * ld + vc0) * 2u, vbase, dv__);                                                        \
    if (DPW == 2) DMA16((uint32_t)(g1__ * ld + vc1) * 2u, vbase, dv__ + 1024);                                   \
  } while (0)
  // full items: tile kt_ = keys 64 kt_ .. 64 kt_ + 63 in tile-row order.  The tile is a SCALAR base (kt_ is a loop counter on the
  // scalar unit), the lane's share of it the constant offsets dK0 .. dV1: nothing is computed on the vector unit per tile, except
  // in the last tile of a sequence whose rows past the end are clamped (wave-uniform branch).
  const uint32_t dK0 = (uint32_t)(drow0 * ld + kc0) * 2u, dK1 = (uint32_t)(drow1 * ld + kc1) * 2u;
  const uint32_t dV0 = (uint32_t)(drow0 * ld + vc0) * 2u, dV1 = (uint32_t)(drow1 * ld + vc1) * 2u;
#define DMA_TILE(kt_, buf_)                                                                                      \
  do {                                                                                                           \
    const int key0__ = (kt_) * KB;                                                                               \
    const bf16_t* kb__ = kbase + (size_t)key0__ * ld;                                                            \
    const bf16_t* vb__ = vbase + (size_t)key0__ * ld;                                                            \
    const uint32_t dk__ = kl_lds + (buf_) * (KB * 128) + wave_s * (1024 * DPW);                                  \
    const uint32_t dv__ = vl_lds + (buf_) * (KB * 128) + wave_s * (1024 * DPW);                                  \
    if (key0__ + KB <= S) {                                                                                      \
      unsigned keep__;                                                                                           \
      if (DPW == 2)                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"        \
                     "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"                            \
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"                            \
                     "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\ts_mov_b32 m0, %0"           \
                     : "=&s"(keep__) : "v"(dK0), "v"(dK1), "v"(dV0), "v"(dV1), "s"(kb__), "s"(vb__), "s"(dk__),       \
                       "s"(dk__ + 1024), "s"(dv__), "s"(dv__ + 1024) : "memory");                                   \
      else                                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"        \
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\ts_mov_b32 m0, %0"            \
                     : "=&s"(keep__) : "v"(dK0), "v"(dV0), "s"(kb__), "s"(vb__), "s"(dk__), "s"(dv__) : "memory");    \
    } else {                                                                                                     \
      const int lim__ = S - 1 - key0__;                                                                          \
      const int g0__ = drow0 < lim__ ? drow0 : lim__, g1__ = drow1 < lim__ ? drow1 : lim__;                      \
      DMA16((uint32_t)(g0__ * ld + kc0) * 2u, kb__, dk__);                                                       \
      if (DPW == 2) DMA16((uint32_t)(g1__ * ld + kc1) * 2u, kb__, dk__ + 1024);                                  \
      DMA16((uint32_t)(g0__ * ld + vc0) * 2u, vb__, dv__);                                                       \
      if (DPW == 2) DMA16((uint32_t)(g1__ * ld + vc1) * 2u, vb__, dv__ + 1024);                                  \
    }                                                                                                            \
  } while (0)

  f32x16 o_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  if (PRE) c_exp = 1.0f;     // the scores are exponents already

  // ---- lane-constant LDS byte offsets, hoisted out of the key loop (all per-tile variation is an immediate) ----
  // K fragment (A operand of S^T): key row t*32 + r, 16-byte chunk (2ks + h) ^ ((row>>1)&7); (row>>1)&7 == (r>>1)&7
  const int ksw = (r >> 1) & 7;
  const int koff0 = r * 128 + (((0 * 2 + h) ^ ksw) << 4);
  const int koff1 = r * 128 + (((1 * 2 + h) ^ ksw) << 4);
  const int koff2 = r * 128 + (((2 * 2 + h) ^ ksw) << 4);
  const int koff3 = r * 128 + (((3 * 2 + h) ^ ksw) << 4);
  // V^T fragment via ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its block;
  // rows 4h + tq (+16sp + 32t), 64-byte half (dt ^ ((row>>1)&1)), (row>>1)&1 == (tq>>1)&1
  const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
  const int vsw = (tq >> 1) & 1;
  const int vlane = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
  const int voff_d0 = vlane + (vsw ? 64 : 0);   // dt = 0
  const int voff_d1 = vlane + (vsw ? 0 : 64);   // dt = 1
  const char* const kbase_lds = reinterpret_cast<const char*>(&kl[0][0]);
  const char* const vbase_lds = reinterpret_cast<const char*>(&vl[0][0]);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + KB - 1) / KB;
  constexpr bool LAZY = PRE && ATTN_LAZY;
  TL_MARK(1);
  if (mode == 0) {
  // PRE: start vectors of the two score accumulators = -m_run per lane (query); one per chain, so that neither MFMA chain has
  // to copy its start vector into its accumulator first
  f32x16 negm0 = zero16;
  // third O^T tile: its V^T operand is the constant row A[0][k] = 1 (lanes 0 and 32), so its row 0 accumulates the row sums
  // ATTN_MSUM (opt-in build flag): row sums on the matrix pipe instead of 32 v_add_f32 per tile.  Measured neutral to -1 % end to
  // end (28.1 k vs 28.4 k clips/s): the 4 extra MFMAs per tile load the pipe the waves of a SIMD share as much as the adds load
  // the issue port, and cost 9 more registers.  Never for the 8-wave paired kernel (held to 128 registers).
#ifdef ATTN_MSUM
  constexpr bool MSUM = NE == 1;
#else
  constexpr bool MSUM = false;
#endif
  f32x16 o_sum = zero16;
  const bf16_t one_or_zero = (bf16_t)(r == 0 ? 1.0f : 0.0f);
  const bf16x8 ones_frag = {one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero};
  if (PRE) m_run = 0.f;            // placeholder until the first tile sets the reference
  DMA_TILE(0, 0);
  STAMP_DECL;
  STAMP_START();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of tile kt has landed
    STAMP(0);                             // segment 0: wait for this wave's own DMA
    __syncthreads();                      // tile complete; every wave is done with the other stage
    STAMP(1);                             // segment 1: barrier
    if (!ATTN_DMA_LATE && kt + 1 < nkt) DMA_TILE(kt + 1, buf ^ 1);
    const char* kt_lds = kbase_lds + buf * (KB * 128);
    const char* vt_lds = vbase_lds + buf * (KB * 128);

    // ---- S^T = K Q^T (64 keys x 32 queries per wave) ----
    // all 8 K fragments of the tile are requested before the first MFMA (left to itself the compiler issues each read right
    // in front of its MFMA and waits for it: 8 exposed LDS latencies per tile)
    f32x16 s_acc[2];
    auto scores = [&]() {
      bf16x8 kf[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        kf[t][0] = *reinterpret_cast<const bf16x8*>(kt_lds + koff0 + t * 4096);
        kf[t][1] = *reinterpret_cast<const bf16x8*>(kt_lds + koff1 + t * 4096);
        kf[t][2] = *reinterpret_cast<const bf16x8*>(kt_lds + koff2 + t * 4096);
        kf[t][3] = *reinterpret_cast<const bf16x8*>(kt_lds + koff3 + t * 4096);
      }
      // the two score tiles interleaved: consecutive MFMAs are independent, each chain of 4 has a tile's worth of slack
      if (PRE) {
        // both chains start from the same -m vector and must leave it intact: written as asm with early-clobber outputs, so the
        // destination is a fresh register range (the builtin form makes hipcc copy the 16 registers and accumulate in place).
        // Operands come from LDS reads (waited for by the compiler) or are long-lived: no VALU-write hazard ahead of them;
        // the consumers are the next MFMAs of the same chain (accumulate dependency, interlocked in hardware).
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s_acc[0]) : "v"(kf[0][0]), "v"(qf[0]), "v"(negm0));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s_acc[1]) : "v"(kf[1][0]), "v"(qf[0]), "v"(negm0));
      } else {
        s_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][0], qf[0], zero16, 0, 0, 0);
        s_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][0], qf[0], zero16, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 1; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          s_acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t][ks], qf[ks], s_acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, PRE ? 6 : 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      // mask keys past the end of the sequence (last tile only)
      if (kt * KB + KB > S) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = kt * KB + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (key >= S) s_acc[t][e] = -INFINITY;
          }
      }
    };
    // row maximum of the tile's scores: four independent chains (a single chain of 16 dependent v_max3 is pure latency for a wave
    // that has nothing else to issue), then the two lane halves of a query exchange their maxima with one v_permlane32_swap
    auto row_max = [&]() -> float {
      float c4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = c >> 1, e0 = (c & 1) * 8;
        float a = fmaxf(fmaxf(s_acc[t][e0], s_acc[t][e0 + 1]), s_acc[t][e0 + 2]);
        a = fmaxf(fmaxf(a, s_acc[t][e0 + 3]), s_acc[t][e0 + 4]);
        a = fmaxf(fmaxf(a, s_acc[t][e0 + 5]), s_acc[t][e0 + 6]);
        c4[c] = fmaxf(a, s_acc[t][e0 + 7]);
      }
      const float m2 = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
      const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2), __builtin_bit_cast(unsigned, m2), false, false);
      // (elements through named unsigneds: __builtin_bit_cast applied to the subscript of the builtin's vector result reads element 0
      // for BOTH - hipcc 7.2 - and the maximum silently became the lower lane half's alone: found in round 5 by spiking keys of the
      // upper half, test_attention_swp_reference_shift_branch / test_attention_online_softmax_rescale_branch)
      const unsigned u0 = sw[0], u1 = sw[1];
      return fmaxf(__uint_as_float(u0), __uint_as_float(u1));
    };
    // PRE: move the running reference up to the tile's maximum mx (relative to the current reference): scores, sums, O and the start
    // vector of the score accumulators shift with it
    auto shift_reference = [&](const float mx) {
      const float d = kt == 0 ? mx - m_run : fmaxf(mx, 0.f);      // first tile: m_run is the placeholder 0
      const float alpha = kt == 0 ? 1.f : __builtin_amdgcn_exp2f(-d);   // nothing accumulated yet on the first tile
      if (MSUM) o_sum[0] *= alpha; else l_run *= alpha;           // rows 1.. of the third tile are zero
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s_acc[t][e] -= d;
      m_run += d;
#pragma unroll
      for (int e = 0; e < 16; ++e) negm0[e] = -m_run;
    };
    // p = exp2(score) in place; returns the lane's sum over its 32 keys.  Two packed chains: v_pk_add_f32 takes two p per
    // instruction (16 instead of 32 adds per tile)
    auto exp_and_sum = [&]() -> float {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s_acc[t][e] = __builtin_amdgcn_exp2f(s_acc[t][e]);
      f32x2 ps0 = {0.f, 0.f}, ps1 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          ps0 += (f32x2){s_acc[t][e], s_acc[t][e + 1]};
          ps1 += (f32x2){s_acc[t][e + 2], s_acc[t][e + 3]};
        }
      ps0 += ps1;
      return ps0[0] + ps0[1];
    };
    scores();
    // the next tile's DMA is issued BEHIND the score MFMAs: its four instructions take a wave 250-480 cycles to get accepted (one
    // address path per CU, 64 bytes a cycle: tools/ubench/valu_rates.hip kind 22, profiles/r03_attn_stamps.txt) - time in which the
    // MFMAs already run instead of waiting behind it
    if (ATTN_DMA_LATE && kt + 1 < nkt) DMA_TILE(kt + 1, buf ^ 1);
    if (ATTN_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    if (ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    STAMP(2);                             // segment 2: DMA issue, K fragment reads, S MFMA issue
    // ---- online softmax (fp32) ----
    float psum;
    if (PRE) {
      // the tile's scores are relative to the running reference m_run (they started from -m_run).  The reference only has to keep
      // exp2(score - m_run) in range, it need not be the exact maximum (bf16 P keeps its 8 significant bits at any magnitude, sums
      // are fp32).  With the exact maximum as reference, one of a wave's 32 queries meets a new maximum in most tiles
      // (1 - (1 - 1/(kt+1))^32) and the whole wave pays the 80-instruction shift of scores, sums and accumulators nearly every tile.
      if (LAZY) {
        // no row maximum per tile at all: the scores are exponentiated against the reference as it stands, and the row sums - needed
        // anyway - tell whether that was safe.  A lane whose 32 p's sum to more than 2^ATTN_LAZY_LOG2 (or to inf / NaN) sends the wave
        // through the tile again (K is still in LDS), this time with the exact maximum and the shift.  Only an entry's first tile
        // (no reference yet) and such tiles pay the ~26 vector instructions of the maximum.
        // (one site for the shift: a second one makes hipcc carry the 16-register start vector through copies on the hot path)
        bool exact = kt == 0;
        if (!exact) {
          psum = exp_and_sum();
          exact = __builtin_amdgcn_ballot_w64(!(psum <= ATTN_LAZY_BIG)) != 0ull;
          if (exact) scores();
        }
        STAMP(3);
        if (exact) {
          shift_reference(row_max());
          psum = exp_and_sum();
        }
      } else {
        // the reference is moved up only when some score of the tile exceeds it by more than defer_thr (p <= 2^thr)
        const float mx = row_max();
        STAMP(3);                           // segment 3: S MFMA completion, row maximum, lane exchange
        if (kt == 0 || __builtin_amdgcn_ballot_w64(mx > defer_thr) != 0ull) shift_reference(mx);
        psum = exp_and_sum();
      }
    } else {
      const float mx = row_max();
      STAMP(3);
      const float m_new = (mx - m_run) * c_exp > defer_thr ? mx : m_run;   // deferred reference (see the PRE branch); the -inf start moves
      const float mc = m_new * c_exp;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s_acc[t][e] = fmaf(s_acc[t][e], c_exp, -mc);
      psum = exp_and_sum();
      // rescale the running state only when some row's reference moved (wave-uniform branch; exact: alpha == 1 otherwise)
      if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
        if (MSUM) o_sum[0] *= alpha; else l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
        m_run = m_new;
      }
    }
    if (!MSUM) l_run += psum;
    // P as bf16 B-operand fragments: k-step (t, sp) = registers 8sp..8sp+7 of score tile t
#define PFRAG(t_, sp_)                                                                                             \
  ((bf16x8){(bf16_t)s_acc[t_][8 * sp_ + 0], (bf16_t)s_acc[t_][8 * sp_ + 1], (bf16_t)s_acc[t_][8 * sp_ + 2],        \
            (bf16_t)s_acc[t_][8 * sp_ + 3], (bf16_t)s_acc[t_][8 * sp_ + 4], (bf16_t)s_acc[t_][8 * sp_ + 5],        \
            (bf16_t)s_acc[t_][8 * sp_ + 6], (bf16_t)s_acc[t_][8 * sp_ + 7]})
    const bf16x8 pf00 = PFRAG(0, 0), pf01 = PFRAG(0, 1), pf10 = PFRAG(1, 0), pf11 = PFRAG(1, 1);
#undef PFRAG
#ifdef ATTN_STAMPS
    asm volatile("" ::"v"(pf00), "v"(pf01), "v"(pf10), "v"(pf11));
#endif
    if (ATTN_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(2);
    STAMP(4);                             // segment 4: exp2, bf16 pack

    // ---- O^T += V^T P^T, row sums += 1^T P^T ----
    // the V^T fragments of a d-half are requested together, one half ahead of the MFMAs that use them
#define VFRAG(dt_, t_, sp_)                                                                                        \
  ({                                                                                                               \
    const char* vb__ = vt_lds + ((dt_) == 0 ? voff_d0 : voff_d1) + (t_) * 4096 + (sp_) * 2048;                     \
    const bf16x4 lo__ = lds_read_tr16(vb__), hi__ = lds_read_tr16(vb__ + 1024);                                    \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                              \
  })
    {
      bf16x8 vf0[4], vf1[4];
      vf0[0] = VFRAG(0, 0, 0); vf0[1] = VFRAG(0, 0, 1); vf0[2] = VFRAG(0, 1, 0); vf0[3] = VFRAG(0, 1, 1);
      vf1[0] = VFRAG(1, 0, 0); vf1[1] = VFRAG(1, 0, 1); vf1[2] = VFRAG(1, 1, 0); vf1[3] = VFRAG(1, 1, 1);
      if (MSUM) o_sum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_frag, pf00, o_sum, 0, 0, 0);
      o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[0], pf00, o_acc[0], 0, 0, 0);
      o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[0], pf00, o_acc[1], 0, 0, 0);
      if (MSUM) o_sum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_frag, pf01, o_sum, 0, 0, 0);
      o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[1], pf01, o_acc[0], 0, 0, 0);
      o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[1], pf01, o_acc[1], 0, 0, 0);
      if (MSUM) o_sum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_frag, pf10, o_sum, 0, 0, 0);
      o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[2], pf10, o_acc[0], 0, 0, 0);
      o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[2], pf10, o_acc[1], 0, 0, 0);
      if (MSUM) o_sum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_frag, pf11, o_sum, 0, 0, 0);
      o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[3], pf11, o_acc[0], 0, 0, 0);
      o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[3], pf11, o_acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MSUM ? 12 : 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef VFRAG
    STAMP(5);                             // segment 5: V fragment reads, PV MFMA issue

  }
#ifdef ATTN_STAMPS
  const unsigned long long st_loop_end__ = st_prev__;
  // [0..5] loop segments of a full item, [6] its number of key tiles, [7] unused (round 2 shipped this build without the write-out:
  // profiles/r02_attn_stamps.txt held zeros for this kernel)
  if (stamps && blockIdx.x % 37 == 0 && lane == 0) {
    long long* dst = stamps + ((size_t)(blockIdx.x / 37) * 4 * NE + w8) * 8;
    for (int i = 0; i < 6; ++i) dst[i] = (long long)st_acc__[i];
    dst[6] = (long long)nkt;
    dst[7] = (long long)(st_loop_end__ & 0);
  }
#endif
  if (MSUM) l_run = o_sum[0];      // row 0 of the third tile (lanes 0..31); lanes 32..63 hold its row 4 = 0
  } else {
    // ================= half item: 64 queries, the key range split between the two wave pairs =================
    const int kh = __builtin_amdgcn_readfirstlane(wave >> 1);
    const int n32 = (S + 31) >> 5, n_half = (n32 + 1) >> 1;      // 32-key tiles; pair 0: [0, n_half), pair 1: [n_half, n32)
    DMA2(0, n_half * 32, 0);
    for (int st = 0; st < n_half; ++st) {
      const int buf = st & 1;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (st + 1 < n_half) DMA2((st + 1) * 32, (n_half + st + 1) * 32, buf ^ 1);
      const int key0 = (kh ? n_half + st : st) * 32;
      if (key0 < S) {   // wave-uniform; pair 1 may run out of tiles one step early
        // rows 32 kh .. 32 kh + 31 of the staged tile are this pair's keys
        const char* kt_lds = kbase_lds + buf * (KB * 128) + kh * 4096;
        const char* vt_lds = vbase_lds + buf * (KB * 128) + kh * 4096;
        f32x16 sc;
        {
          bf16x8 kf[4];
          kf[0] = *reinterpret_cast<const bf16x8*>(kt_lds + koff0);
          kf[1] = *reinterpret_cast<const bf16x8*>(kt_lds + koff1);
          kf[2] = *reinterpret_cast<const bf16x8*>(kt_lds + koff2);
          kf[3] = *reinterpret_cast<const bf16x8*>(kt_lds + koff3);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks ? sc : zero16, 0, 0, 0);
        }
        if (key0 + 32 > S) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (key >= S) sc[e] = -INFINITY;
          }
        }
        float mx = sc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sc[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = (mx - m_run) * c_exp > defer_thr ? mx : m_run;
        const float mc = m_new * c_exp;
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          sc[e] = __builtin_amdgcn_exp2f(fmaf(sc[e], c_exp, -mc));
          psum += sc[e];
        }
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
          l_run *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
          m_run = m_new;
        }
        l_run += psum;
#define PFRAGH(sp_)                                                                                                \
  ((bf16x8){(bf16_t)sc[8 * sp_ + 0], (bf16_t)sc[8 * sp_ + 1], (bf16_t)sc[8 * sp_ + 2], (bf16_t)sc[8 * sp_ + 3],      \
            (bf16_t)sc[8 * sp_ + 4], (bf16_t)sc[8 * sp_ + 5], (bf16_t)sc[8 * sp_ + 6], (bf16_t)sc[8 * sp_ + 7]})
        const bf16x8 pf0 = PFRAGH(0), pf1 = PFRAGH(1);
#undef PFRAGH
#define VFRAGH(dt_, sp_)                                                                                           \
  ({                                                                                                               \
    const char* vb__ = vt_lds + ((dt_) == 0 ? voff_d0 : voff_d1) + (sp_) * 2048;                                   \
    const bf16x4 lo__ = lds_read_tr16(vb__), hi__ = lds_read_tr16(vb__ + 1024);                                    \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                              \
  })
        const bf16x8 v00 = VFRAGH(0, 0), v01 = VFRAGH(0, 1), v10 = VFRAGH(1, 0), v11 = VFRAGH(1, 1);
#undef VFRAGH
        o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v00, pf0, o_acc[0], 0, 0, 0);
        o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v10, pf0, o_acc[1], 0, 0, 0);
        o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v01, pf1, o_acc[0], 0, 0, 0);
        o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v11, pf1, o_acc[1], 0, 0, 0);
      }
    }
    __syncthreads();   // every wave is done with the tiles: they become the exchange buffer
    // merge the two key halves: pair 1 hands its state to pair 0 (same query rows, same lane layout) through the K / V tiles
    // O state: 16 KB per entry ([qg][8][64 lanes] float4) - entry 0 in the K tiles, entry 1 in the V tiles; (m, l) in xm_s
    f32x4* xo = reinterpret_cast<f32x4*>(ent ? &vl[0][0] : &kl[0][0]);
    float* xm = xm_s + ent * 256;                         // [qg][m | l][64 lanes]
    const int qg = wave & 1;
    if (kh == 1) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          xo[(qg * 8 + dt * 4 + g) * 64 + lane] = (f32x4){o_acc[dt][4 * g], o_acc[dt][4 * g + 1], o_acc[dt][4 * g + 2], o_acc[dt][4 * g + 3]};
      xm[(qg * 2 + 0) * 64 + lane] = m_run;
      xm[(qg * 2 + 1) * 64 + lane] = l_run;
    }
    __syncthreads();
    if (kh == 1) return;
    {
      const float m1 = xm[(qg * 2 + 0) * 64 + lane], l1 = xm[(qg * 2 + 1) * 64 + lane];
      const float m_all = fmaxf(m_run, m1);
      const float a0 = __builtin_amdgcn_exp2f((m_run - m_all) * c_exp), a1 = __builtin_amdgcn_exp2f((m1 - m_all) * c_exp);
      l_run = l_run * a0 + l1 * a1;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o1 = xo[(qg * 8 + dt * 4 + g) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) o_acc[dt][4 * g + e] = o_acc[dt][4 * g + e] * a0 + o1[e] * a1;
        }
      m_run = m_all;
    }
  }
#undef DMA2
#undef DMA_TILE
#undef DMA16

  TL_MARK(2);
  // ---- normalise, gate, store: lane holds O[query r][32dt + 8g + 4h + 0..3] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = __builtin_amdgcn_rcpf(l_tot);
  if (!live) return;      // NE == 2 only
  if (TAPE && lse_out && qrow < S && h == 0)   // natural-log LSE of the scaled scores (training tape): scale*max + ln(sum)
    lse_out[(size_t)(s0 + qrow) * (d_model >> 6) + head] = m_run * (c_exp * 0.69314718055994530942f) + __logf(l_tot);
  if constexpr (MXO) {
    uint8_t* const oq = reinterpret_cast<uint8_t*>(out_raw) + (size_t)(s0 + qrc) * d_model + head * 64;
    uint8_t* const omx = reinterpret_cast<uint8_t*>(lse_out) + (size_t)(s0 + qrc) * ldo;
    const int nkp = ldo >> 2;
    const bf16_t* grow = gbase + (size_t)qrc * ld;
    const bool store = qrow < S;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x4 v[4];
      float amax = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        v[g] = (f32x4){o_acc[dt][4 * g] * inv_l, o_acc[dt][4 * g + 1] * inv_l, o_acc[dt][4 * g + 2] * inv_l, o_acc[dt][4 * g + 3] * inv_l};
        if (GATE) {
          const f32x4 gt = Vec4<bf16_t>::load(grow + d0);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g][e] *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt[e] * -1.44269504088896340736f));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[g][e] = round_to<bf16_t>(v[g][e]);            // the value the bf16 kernel stores: the image equals k_quant_mx_fp8 of that output
          amax = fmaxf(amax, fabsf(v[g][e]));
        }
      }
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const uint32_t tb = __float_as_uint(amax * (1.0f / 448.0f));
      int byte = (int)((tb >> 23) & 0xFF) + ((tb & 0x7FFFFF) ? 1 : 0);
      byte = amax > 0.f ? (byte < 1 ? 1 : (byte > 254 ? 254 : byte)) : 127;
      const float inv = __uint_as_float((uint32_t)(254 - byte) << 23);
      int w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int t = 0;
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][0] * inv, v[g][1] * inv, t, false);
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][2] * inv, v[g][3] * inv, t, true);
        w[g] = t;
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        // as the bf16 stores: lane (r, 0) ends with the groups 2gp of both lanes = 8 consecutive features, lane (r, 1) with the groups 2gp + 1
        const auto sx = __builtin_amdgcn_permlane32_swap((unsigned)w[2 * gp], (unsigned)w[2 * gp + 1], false, false);
        if (store) *reinterpret_cast<uint2*>(oq + dt * 32 + 16 * gp + 8 * h) = make_uint2(sx[0], sx[1]);
      }
      if (store && h == 0) {
        const int blk = head * 2 + dt;
        omx[(blk & 3) * nkp + (blk >> 2)] = (uint8_t)byte;
      }
    }
  } else {
    // Every lane computes (rows past the end are clamped for the gate load), lanes exchange, rows < S store.  Lanes (r, 0) and
    // (r, 1) hold the 4-feature groups 8g + 0..3 and 8g + 4..7 of a row: for each pair of groups (g, g + 1) one
    // v_permlane32_swap per dword leaves lane (r, 0) with features 8g .. 8g + 7 and lane (r, 1) with 8(g+1) .. 8(g+1) + 7:
    // one 16-byte store per lane and pair instead of two 8-byte ones.
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
          f32x4 v = {o_acc[dt][4 * g] * inv_l, o_acc[dt][4 * g + 1] * inv_l, o_acc[dt][4 * g + 2] * inv_l, o_acc[dt][4 * g + 3] * inv_l};
          if (TAPE && out_raw) {   // training tape: the ungated output as well; the gate then multiplies the stored (rounded) value
            if (store) Vec4<bf16_t>::store(out_raw + (size_t)(s0 + qrow) * ldo + head * 64 + d0, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = round_to<bf16_t>(v[e]);
          }
          if (GATE) {
            const f32x4 gt = Vec4<bf16_t>::load(grow + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt[e] * -1.44269504088896340736f));
          }
          const bf16x4 b4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          pk[gg] = __builtin_bit_cast(uint2, b4);
        }
        // lanes 32..63 of pk[0] <-> lanes 0..31 of pk[1]
        const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        // lane (r,0): [own group 2gp | partner's group 2gp] = features 16gp .. 16gp + 7 of the d-half;
        // lane (r,1): [partner's group 2gp+1 | own group 2gp+1] = features 16gp + 8 .. 16gp + 15
        const uint4 o16 = {sx[0], sy[0], sx[1], sy[1]};
        if (store) *reinterpret_cast<uint4*>(orow + dt * 32 + 16 * gp + 8 * h) = o16;
      }
  }
#ifdef ATTN_TIMELINE
  __builtin_amdgcn_s_waitcnt(0x0F70);      // the entry's stores have been acknowledged
  TL_MARK(3);
  if (stamps && w8 == 0 && lane == 0) {
    unsigned hw__, xcc__;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw__), "=s"(xcc__));
    long long* dst = stamps + (size_t)tix * 8;
    for (int i = 0; i < 4; ++i) dst[i] = (long long)tl_r__[i];
    dst[4] = (long long)tl_c__[0];
    dst[5] = (long long)tl_c__[3];
    dst[6] = (long long)hw__;
    dst[7] = (long long)xcc__;
  }
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined bf16 kernel for tables of FULL items with pre-scaled q.  OPT-IN (flag TTV_ATTN_PIPE; towers: TTV_ATTN_PIPE=1): correct (same tests as
// k_attn_bf16) but 6-10 % SLOWER than it on MI355X at the benchmark shape (62.8 vs 59.3 us per launch inside bench.py, 67 vs 60 us
// back to back); kept as the measured record of what in-wave pipelining buys on this part - nothing - and why.
//
// Idea.  k_attn_bf16 walks S MFMAs -> row maximum -> exp2 / pack -> PV MFMAs strictly in sequence in every wave.  Here
//   * the phases of neighbouring steps overlap INSIDE the wave.  A step u is a 32-key half of a 64-key tile (two score sets of
//     16 registers; with 64-key steps the loop wants 184 VGPRs and spills at the 168 of three waves per SIMD):
//         step u :  PV(u-1) | first half of softmax(S(u)),  then  S(u+1) = K(u+1) Q^T | second half      (4 + 4 MFMAs)
//     and the LDS fragments of a block are read one block ahead of their MFMAs;
//   * there is NO row maximum on the main path.  The reference m of a row is the maximum of its first 32 scores; later steps
//     compute p = exp2(s - m) against it unchecked and look at the lane's row sum afterwards: only when that exceeds
//     2^ATTN_PIPE_THR (this includes inf / nan) a rare branch takes the step's raw scores again (4 MFMAs, K straight from global
//     memory), moves the reference to their true maximum, rescales O and l, and recomputes the step's p.  Any reference gives
//     the same quotient; bf16 P and fp32 sums have the range for 2^40;
//   * LDS addresses are six loop-invariant registers plus the ring slot's scalar offset plus an immediate, row sums and the
//     reference subtraction use the packed fp32 add, the DMA sources are running 64-bit bases (one asm statement per tile).
// Tile t of K is read in steps 2t-1 and 2t, tile t of V in steps 2t+1 and 2t+2: ONE barrier per tile, at the top of the odd step
// 2t+1, where K(t+1) and V(t) must have landed and the slots of K(t) and V(t-1) fall free.  Three-slot rings, filled two tile
// periods ahead by LDS-DMA (K(t+3), V(t+2) issued there; tile indices past the end re-fetch the last tile so that every tile issues
// the same four DMA instructions per wave and the counted wait `s_waitcnt vmcnt(4)` is a constant); raw s_barrier, because a
// __syncthreads() would drain the DMA in flight.  48 KB LDS, 141 VGPRs, three blocks per CU.
//
// Why it does not pay ("attention: where the time goes"; tools/ubench/valu_rates.hip, tools/attn_knockout.sh):
//   * on one SIMD, matrix and vector work barely overlap, neither across waves nor inside one: 16 MFMAs + 96 plain + 32 exp2
//     instructions per iteration cost 584 cycles per wave at three waves per SIMD whether interleaved or in two phases - the sum of
//     the MFMA-only (350) and the vector-only loop (291) is 641.  A wave's phases being serial is therefore not the loss;
//   * knocking ingredients out of this kernel (garbage results, same launch): no in-loop DMA -4 us, no barrier -2 us, no LDS
//     fragment reads -6.5 us, a quarter of the softmax arithmetic -11 us, all four 65 -> 43 us: the costs ADD, nothing hides
//     behind anything else, and ~10 us of every launch are the two rounds' prologues and epilogues.
// ------------------------------------------------------------------------------------------------
#define ATTN_PIPE_THR 40.0f
// knock-out switches of diagnostic builds (tools/attn_knockout.sh): what does the launch cost WITHOUT the in-loop DMA, the
// barrier, three quarters of the exp2 arithmetic?  Results are garbage then; the product build has all three at 0.
#ifndef PP_KO_DMA
#define PP_KO_DMA 0
#endif
#ifndef PP_KO_BARRIER
#define PP_KO_BARRIER 0
#endif
#ifndef PP_KO_LDS
#define PP_KO_LDS 0
#endif
#ifndef PP_KO_VALU
#define PP_KO_VALU 0
#endif
template <bool GATE>
__global__ __launch_bounds__(256, 3) void k_attn_pipe(const bf16_t* __restrict__ qkvg, int ld, bf16_t* __restrict__ out, int ldo,
                                                      const int* __restrict__ cu, const int* __restrict__ qblocks, int d_model, int gqa,
                                                      int rep, float defer_thr, long long* stamps) {
  __shared__ __attribute__((aligned(16))) uint4 kl[3][KB * 8];
  __shared__ __attribute__((aligned(16))) uint4 vl[3][KB * 8];
#ifdef ATTN_STAMPS
  unsigned long long st_entry__;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_entry__)::"memory");
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tix = blockIdx.x;
  const int seq = qblocks[4 * tix], q0 = qblocks[4 * tix + 1], head = qblocks[4 * tix + 2];
  if (seq < 0) return;
  const int s0 = cu[seq], S = cu[seq + 1] - s0;
  const int kvh = head / rep;
  const bf16_t* qbase = qkvg + (size_t)s0 * ld + head * 64;
  const bf16_t* gbase = qkvg + (size_t)s0 * ld + d_model + head * 64;
  const bf16_t* kbase = qkvg + (size_t)s0 * ld + 2 * d_model + kvh * 64;
  const bf16_t* vbase = kbase + gqa;
  const int qrow = q0 + wave * 32 + r;
  const int qrc = qrow < S ? qrow : S - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrc * ld + ks * 16 + h * 8);
  // a use of q ahead of the DMA: the compiler's wait for these loads (it cannot count the DMA of the asm statements, so it waits for
  // vmcnt(0)) lands here and not in front of the first MFMA, where it would drain the whole prologue
  asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));

  // DMA shares: wave w stages tile rows 8 w + (lane >> 3) and that + 32 of K and of V.  Rows 32 apart have the same chunk swizzle
  // (K: (row >> 1) & 7, V: ((row >> 1) & 1) << 2), so ONE per-lane offset serves both instructions of an operand (the 32 rows go
  // into the scalar base) - two VGPRs live through the loop instead of four.
  const uint32_t kl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&kl[0][0];
  const uint32_t vl_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&vl[0][0];
  const int drow = wave * 8 + (lane >> 3);
  const int kc = ((lane & 7) ^ ((drow >> 1) & 7)) * 8;
  const int vc = ((lane & 7) ^ (((drow >> 1) & 1) << 2)) * 8;
  const uint32_t dK = (uint32_t)(drow * ld + kc) * 2u;
  const uint32_t dV = (uint32_t)(drow * ld + vc) * 2u;
  const int nkt = (S + KB - 1) / KB;
#define PP_DMA(voff_, base_, dst_)                                                                               \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
  // one operand tile (two DMA instructions per wave, always): tile index clamped to the last one, rows past the end clamped
#define PP_TILE(base_, lds_, kt_, soff_, c_, d_)                                                                 \
  do {                                                                                                           \
    const int ktc__ = (kt_) < nkt ? (kt_) : nkt - 1;                                                             \
    const int key0__ = ktc__ * KB;                                                                               \
    const bf16_t* b__ = (base_) + (size_t)key0__ * ld;                                                           \
    const uint32_t dst__ = (lds_) + (soff_) + wave * 1024;                                                       \
    if (key0__ + KB <= S) {                                                                                      \
      PP_DMA(d_, b__, dst__);                                                                                    \
      PP_DMA(d_, b__ + (size_t)32 * ld, dst__ + 4096);                                                           \
    } else {                                                                                                     \
      const int lim__ = S - 1 - key0__;                                                                          \
      const int g0__ = drow < lim__ ? drow : lim__, g1__ = drow + 32 < lim__ ? drow + 32 : lim__;                \
      PP_DMA((uint32_t)(g0__ * ld + (c_)) * 2u, b__, dst__);                                                     \
      PP_DMA((uint32_t)(g1__ * ld + (c_)) * 2u, b__, dst__ + 4096);                                              \
    }                                                                                                            \
  } while (0)

  // the four DMA instructions of a tile in ONE statement (M0 saved and restored once), sources = running 64-bit bases: tiles
  // before the sequence's last one need no clamping
#define PP_DMA4(kp_, kp32_, kdst_, vp_, vp32_, vdst_)                                                            \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\t"                                                                          \
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"                             \
                 "s_add_u32 m0, %5, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"                     \
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"                             \
                 "s_add_u32 m0, %8, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"                     \
                 "s_mov_b32 m0, %0"                                                                              \
                 : "=&s"(keep__)                                                                                 \
                 : "v"(dK), "v"(dV), "s"(kp_), "s"(kp32_), "s"(kdst_), "s"(vp_), "s"(vp32_), "s"(vdst_)           \
                 : "memory", "scc");                                                                             \
  } while (0)

  // lane-constant LDS addresses (see k_attn_bf16); the ring slot is a scalar byte offset, the half an immediate
  const int ksw = (r >> 1) & 7;
  const char* const kbase_lds = reinterpret_cast<const char*>(&kl[0][0]);
  const char* const vbase_lds = reinterpret_cast<const char*>(&vl[0][0]);
  const char* const ka0 = kbase_lds + r * 128 + (((0 * 2 + h) ^ ksw) << 4);
  const char* const ka1 = kbase_lds + r * 128 + (((1 * 2 + h) ^ ksw) << 4);
  const char* const ka2 = kbase_lds + r * 128 + (((2 * 2 + h) ^ ksw) << 4);
  const char* const ka3 = kbase_lds + r * 128 + (((3 * 2 + h) ^ ksw) << 4);
  const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
  const int vsw = (tq >> 1) & 1;
  const int vlane = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
  const char* const va0 = vbase_lds + vlane + (vsw ? 64 : 0);
  const char* const va1 = vbase_lds + vlane + (vsw ? 0 : 64);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  f32x16 o_acc[2] = {zero16, zero16};
  float m_run = 0.f, l_run = 0.f;
  const float p_limit = __builtin_amdgcn_exp2f(defer_thr);
  bf16x8 pf0, pf1;                    // P(u-1) as B fragments (keys 0-15 and 16-31 of the half)
  f32x16 sA, sB;                      // the two score sets (raw scores)
  bf16x8 kf0, kf1, kf2, kf3;          // K fragments of the next S block, V fragments of the next PV block: read from LDS one
  bf16x8 vf00, vf10, vf01, vf11;      // block ahead of their MFMAs, so that no block opens on an exposed LDS round trip

  // fragment reads of half J_ of the ring slot at byte offset soff_ (J_: literal)
#define PP_LOADK(soff_, J_)                                                                                      \
  do {                                                                                                           \
    if (PP_KO_LDS) { kf0 = qf[0]; kf1 = qf[1]; kf2 = qf[2]; kf3 = qf[3]; break; }                                \
    kf0 = *reinterpret_cast<const bf16x8*>(ka0 + (soff_) + (J_) * 4096);                                         \
    kf1 = *reinterpret_cast<const bf16x8*>(ka1 + (soff_) + (J_) * 4096);                                         \
    kf2 = *reinterpret_cast<const bf16x8*>(ka2 + (soff_) + (J_) * 4096);                                         \
    kf3 = *reinterpret_cast<const bf16x8*>(ka3 + (soff_) + (J_) * 4096);                                         \
  } while (0)
#define PP_VFRAG(va_, OFF_)                                                                                      \
  ({                                                                                                             \
    const bf16x4 lo__ = lds_read_tr16((va_) + (OFF_)), hi__ = lds_read_tr16((va_) + (OFF_) + 1024);              \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                            \
  })
#define PP_LOADV(soff_, J_)                                                                                      \
  do {                                                                                                           \
    if (PP_KO_LDS) { vf00 = vf10 = vf01 = vf11 = qf[0]; break; }                                                 \
    const char* v0__ = va0 + (soff_);                                                                            \
    const char* v1__ = va1 + (soff_);                                                                            \
    vf00 = PP_VFRAG(v0__, (J_) * 4096);                                                                          \
    vf10 = PP_VFRAG(v1__, (J_) * 4096);                                                                          \
    vf01 = PP_VFRAG(v0__, (J_) * 4096 + 2048);                                                                   \
    vf11 = PP_VFRAG(v1__, (J_) * 4096 + 2048);                                                                   \
  } while (0)
  // S = K Q^T from the K fragments in registers
#define PP_SCORES(dst_)                                                                                          \
  do {                                                                                                           \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[0], zero16, 0, 0, 0);                                 \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[1], dst_, 0, 0, 0);                                   \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf2, qf[2], dst_, 0, 0, 0);                                   \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf3, qf[3], dst_, 0, 0, 0);                                   \
  } while (0)
  // O^T += V^T P^T from the V fragments in registers, P in pf0, pf1
#define PP_PV()                                                                                                  \
  do {                                                                                                           \
    o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf00, pf0, o_acc[0], 0, 0, 0);                            \
    o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf10, pf0, o_acc[1], 0, 0, 0);                            \
    o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf01, pf1, o_acc[0], 0, 0, 0);                            \
    o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf11, pf1, o_acc[1], 0, 0, 0);                            \
  } while (0)
  // keys past the end of the sequence (only in the last tile, only when S is not a multiple of 64)
#define PP_MASK(dst_, kt_, J_)                                                                                   \
  do {                                                                                                           \
    if ((kt_) == nkt - 1 && nkt * KB > S) {                                                                      \
      _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) {                                                     \
        const int key__ = (kt_) * KB + (J_) * 32 + (e__ & 3) + 8 * (e__ >> 2) + 4 * h;                           \
        if (key__ >= S) dst_[e__] = -INFINITY;                                                                   \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
  // row maximum of a score set over both lane halves
#define PP_ROWMAX(c_)                                                                                            \
  ({                                                                                                             \
    float a__ = fmaxf(fmaxf(c_[0], c_[1]), c_[2]), b__ = fmaxf(fmaxf(c_[8], c_[9]), c_[10]);                     \
    a__ = fmaxf(fmaxf(a__, c_[3]), c_[4]);                                                                       \
    b__ = fmaxf(fmaxf(b__, c_[11]), c_[12]);                                                                     \
    a__ = fmaxf(fmaxf(a__, c_[5]), c_[6]);                                                                       \
    b__ = fmaxf(fmaxf(b__, c_[13]), c_[14]);                                                                     \
    const float m2__ = fmaxf(fmaxf(a__, c_[7]), fmaxf(b__, c_[15]));                                             \
    const auto sw__ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2__), __builtin_bit_cast(unsigned, m2__), false, false); \
    const unsigned u0__ = sw__[0], u1__ = sw__[1];     /* not __builtin_bit_cast(float, sw__[1]): see k_attn_bf16's row_max */ \
    fmaxf(__uint_as_float(u0__), __uint_as_float(u1__));                                                         \
  })
  // in place p = exp2(s - m) of elements 8 SP_ .. 8 SP_ + 7 of a score set, their sum added to acc_ (two partial sums), and P as
  // a bf16 B fragment -> pa_
#define PP_SOFTMAX8(c_, SP_, pa_, acc_)                                                                          \
  do {                                                                                                           \
    const f32x2 nm__ = {-m_run, -m_run};                                                                         \
    _Pragma("unroll") for (int e__ = 8 * (SP_); e__ < 8 * (SP_) + (PP_KO_VALU ? 2 : 8); e__ += 2) {              \
      const f32x2 x__ = (f32x2){c_[e__], c_[e__ + 1]} + nm__;                                                    \
      c_[e__] = __builtin_amdgcn_exp2f(x__[0]);                                                                  \
      c_[e__ + 1] = __builtin_amdgcn_exp2f(x__[1]);                                                              \
      acc_ += (f32x2){c_[e__], c_[e__ + 1]};                                                                     \
    }                                                                                                            \
    pa_ = (bf16x8){(bf16_t)c_[8 * (SP_) + 0], (bf16_t)c_[8 * (SP_) + 1], (bf16_t)c_[8 * (SP_) + 2], (bf16_t)c_[8 * (SP_) + 3], \
                   (bf16_t)c_[8 * (SP_) + 4], (bf16_t)c_[8 * (SP_) + 5], (bf16_t)c_[8 * (SP_) + 6], (bf16_t)c_[8 * (SP_) + 7]}; \
  } while (0)
  // rare path: the raw scores of half J_ of tile kt_ once more, K fragments straight from global memory (the exp2 ran in place,
  // and for odd steps the tile's ring slot is already being refilled)
#define PP_SCORES_AGAIN(dst_, kt_, J_)                                                                           \
  do {                                                                                                           \
    const int key__ = (kt_) * KB + (J_) * 32 + r;                                                                \
    const bf16_t* kr__ = kbase + (size_t)(key__ < S ? key__ : S - 1) * ld + h * 8;                               \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr__), qf[0], zero16, 0, 0, 0); \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr__ + 16), qf[1], dst_, 0, 0, 0); \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr__ + 32), qf[2], dst_, 0, 0, 0); \
    dst_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr__ + 48), qf[3], dst_, 0, 0, 0); \
    PP_MASK(dst_, kt_, J_);                                                                                      \
  } while (0)

  // One step.  cur_ holds S(u) = scores of half CJ_ of tile ckt_; nxt_ receives S(u+1) (K fragments already in registers, tile
  // nkt_ half NJ_); PV(u-1) uses the V fragments in registers.  PRE1_ / PRE2_: the fragment reads for later blocks, issued at the
  // head of the first / second block.  Each block (4 MFMAs beside half of the softmax arithmetic of S(u)) is ONE basic block.
  // P(u) and the row sum are pinned to it by the empty asm (they are consumed a step later: the compiler would sink the whole
  // exp2 / pack sequence there, away from the MFMAs it is meant to run beside).
#define PP_STEP(cur_, nxt_, FIRST_, PRE1_, PRE2_, NJ_, nkt_, ckt_, CJ_, SEG1_, SEG2_)                            \
  do {                                                                                                           \
    bf16x8 qa__, qb__;                                                                                           \
    f32x2 acc__ = {0.f, 0.f};                                                                                    \
    /* PV(u-1) beside the first 16 keys of the step */                                                           \
    PRE1_;                                                                                                       \
    if (!(FIRST_)) PP_PV();                                                                                      \
    PP_SOFTMAX8(cur_, 0, qa__, acc__);                                                                           \
    asm volatile("" : "+v"(qa__), "+v"(acc__));                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                        \
      if (!(FIRST_)) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);                                                         \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    STAMP(SEG1_);                                                                                                \
    /* S(u+1) beside the other 16 keys */                                                                        \
    PRE2_;                                                                                                       \
    PP_SCORES(nxt_);                                                                                             \
    PP_SOFTMAX8(cur_, 1, qb__, acc__);                                                                           \
    float ps__ = acc__[0] + acc__[1];                                                                            \
    asm volatile("" : "+v"(qb__), "+v"(ps__));                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
      __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);                                                         \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    /* rare: a row sum beyond 2^defer_thr (or inf / nan): take the step's raw scores again, move the reference to their true */ \
    /* maximum, rescale what was accumulated against the old one (PV(u-1) included) and redo the step's p                  */ \
    if (__builtin_amdgcn_ballot_w64(!(ps__ <= p_limit)) != 0ull) {                                               \
      PP_SCORES_AGAIN(cur_, ckt_, CJ_);                                                                          \
      const float d__ = fmaxf(PP_ROWMAX(cur_) - m_run, 0.f);                                                     \
      const float alpha__ = __builtin_amdgcn_exp2f(-d__);                                                        \
      l_run *= alpha__;                                                                                          \
      _Pragma("unroll") for (int dt__ = 0; dt__ < 2; ++dt__)                                                     \
        _Pragma("unroll") for (int e__ = 0; e__ < 16; ++e__) o_acc[dt__][e__] *= alpha__;                        \
      m_run += d__;                                                                                              \
      acc__ = (f32x2){0.f, 0.f};                                                                                 \
      PP_SOFTMAX8(cur_, 0, qa__, acc__);                                                                         \
      PP_SOFTMAX8(cur_, 1, qb__, acc__);                                                                         \
      ps__ = acc__[0] + acc__[1];                                                                                \
    }                                                                                                            \
    pf0 = qa__;                                                                                                  \
    pf1 = qb__;                                                                                                  \
    l_run += ps__;                                                                                               \
    PP_MASK(nxt_, nkt_, NJ_);                                                                                    \
    STAMP(SEG2_);                                                                                                \
  } while (0)
  // One tile t; o0, o1, o2: byte offsets of ring slots t % 3, (t+1) % 3, (t+2) % 3.  Entering: S(2t) in sA, P(2t-1) in pf, the
  // fragments of V(t-1) half 1 in vf.  Steps 2t and 2t+1 with the tile's barrier / DMA hand-over between them:
  //   block A1  reads K(t) half 1      | PV(2t-1)  beside softmax of S(2t)   keys 0-15
  //   block A2                         | S(2t+1)   beside softmax of S(2t)   keys 16-31
  //   wait, barrier, reads V(t) half 0 and K(t+1) half 0, DMA issue (its scalar work hides the LDS round trip)
  //   block B1                         | PV(2t)    beside softmax of S(2t+1) keys 0-15
  //   block B2  reads V(t) half 1      | S(2t+2)   beside softmax of S(2t+1) keys 16-31
#define PP_TILE_STEPS(t_, FIRST_)                                                                                \
  do {                                                                                                           \
    const int tt__ = (t_);                                                                                       \
    PP_STEP(sA, sB, FIRST_, PP_LOADK(o0, 1), (void)0, 1, tt__, tt__, 0, 0, 1);                                   \
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     /* K(t+1), V(t) landed (own share); K(t+2), V(t+1) may fly */ \
    STAMP(2);                                                                                                    \
    if (!PP_KO_BARRIER) __builtin_amdgcn_s_barrier();    /* ... for every wave; all are done with K(t) and V(t-1) */ \
    STAMP(3);                                                                                                    \
    PP_LOADV(o0, 0);                                                                                             \
    PP_LOADK(o1, 0);                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (PP_KO_DMA) {                                                                                             \
    } else if (tt__ + 3 < nkt - 1) {                     /* K(t+3) -> slot of K(t), V(t+2) -> slot of V(t-1) */    \
      const bf16_t* kp32__ = kdma + (size_t)32 * ld;                                                             \
      const bf16_t* vp32__ = vdma + (size_t)32 * ld;                                                             \
      PP_DMA4(kdma, kp32__, kl_lds + o0 + wave * 1024, vdma, vp32__, vl_lds + o2 + wave * 1024);                 \
    } else {                                                                                                     \
      PP_TILE(kbase, kl_lds, tt__ + 3, o0, kc, dK);                                                              \
      PP_TILE(vbase, vl_lds, tt__ + 2, o2, vc, dV);                                                              \
    }                                                                                                            \
    kdma += (size_t)KB * ld;                                                                                     \
    vdma += (size_t)KB * ld;                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    STAMP(4);                                                                                                    \
    PP_STEP(sB, sA, false, (void)0, PP_LOADV(o0, 1), 0, tt__ + 1, tt__, 1, 5, 5);                                \
  } while (0)

  // ---- prologue: K(0) | V(0), K(1) | K(2), V(1) in flight (10 DMA instructions per wave); S(0) and its row maximum ----
  PP_TILE(kbase, kl_lds, 0, 0, kc, dK);
  PP_TILE(vbase, vl_lds, 0, 0, vc, dV);
  PP_TILE(kbase, kl_lds, 1, KB * 128, kc, dK);
  PP_TILE(kbase, kl_lds, 2, 2 * KB * 128, kc, dK);
  PP_TILE(vbase, vl_lds, 1, KB * 128, vc, dV);
  const bf16_t* kdma = kbase + (size_t)3 * KB * ld;      // K(t+3), V(t+2) for t = 0
  const bf16_t* vdma = vbase + (size_t)2 * KB * ld;
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // K(0): this wave's share
  __builtin_amdgcn_s_barrier();
  PP_LOADK(0, 0);
  PP_SCORES(sA);
  PP_MASK(sA, 0, 0);
  m_run = PP_ROWMAX(sA);
  STAMP_DECL;
  STAMP_START();
#ifdef ATTN_STAMPS
  const unsigned long long st_loop_start__ = st_prev__;
#endif
  {
    int o0 = 0, o1 = KB * 128, o2 = 2 * KB * 128;      // ring slot byte offsets, rotated per tile (scalar registers)
    PP_TILE_STEPS(0, true);
    for (int t = 1; t < nkt; ++t) {
      const int o = o0;
      o0 = o1;
      o1 = o2;
      o2 = o;
      PP_TILE_STEPS(t, false);
    }
  }
  // ---- drain: PV(2 nkt - 1), half 1 of V(nkt-1): its fragments were read in the last block; sA holds scores of the re-fetched
  // tile past the end: unused ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_PV();
#ifdef ATTN_STAMPS
  const unsigned long long st_loop_end__ = st_prev__;
#endif
#undef PP_TILE_STEPS
#undef PP_STEP
#undef PP_SCORES_AGAIN
#undef PP_SOFTMAX8
#undef PP_ROWMAX
#undef PP_MASK
#undef PP_PV
#undef PP_SCORES
#undef PP_LOADV
#undef PP_VFRAG
#undef PP_LOADK
#undef PP_DMA4
#undef PP_TILE
#undef PP_DMA

  // ---- normalise, gate, store (as k_attn_bf16) ----
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
          f32x4 v = {o_acc[dt][4 * g] * inv_l, o_acc[dt][4 * g + 1] * inv_l, o_acc[dt][4 * g + 2] * inv_l, o_acc[dt][4 * g + 3] * inv_l};
          if (GATE) {
            const f32x4 gt = Vec4<bf16_t>::load(grow + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt[e] * -1.44269504088896340736f));
          }
          const bf16x4 b4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          pk[gg] = __builtin_bit_cast(uint2, b4);
        }
        const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const uint4 o16 = {sx[0], sy[0], sx[1], sy[1]};
        if (store) *reinterpret_cast<uint4*>(orow + dt * 32 + 16 * gp + 8 * h) = o16;
      }
  }
#ifdef ATTN_STAMPS
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

// ------------------------------------------------------------------------------------------------
// fp32 kernel: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate = an fmaf chain per output; 157 TFLOP/s peak).
// The compute path of `dtype=float32` towers - the mode whose token indices equal the reference's fp32 result.
// Same decomposition as the bf16 kernel: block = 4 waves = 128 query rows of one (sequence, q-head), 32 queries per wave,
// 64-key K / V tiles in LDS (register-staged, the next tile's loads in flight behind the MFMAs of the current one).
//   S^T = K Q^T   : key on the MFMA row, query on the lane.  One MFMA sums 2 head dims: lane half h supplies dims 32h + i at
//                   step i (Q: 32 registers per lane, loaded once; K: ds_read_b128 of the lane's key row, 16-byte chunks
//                   XOR-swizzled by key & 15 so that the 16 lanes of a read group hit 16 different slots).
//   O^T = V^T P^T : the S^T accumulator registers ARE the B operand, unconverted: register e of lane half h is key
//                   (e & 3) + 8 (e >> 2) + 4h of the 32-key sub-tile, so step e sums keys k_e(0), k_e(1) and the A operand is
//                   V[k_e(h)][d = lane & 31 (+32)], a plain ds_read_b32 of the row-major V tile (conflict-free: 32 consecutive
//                   floats per half).
// Softmax in fp32 with exp2 (scale * log2 e folded into one FMA); rescale only when a row maximum moved.
// Half items (mode 1) are computed as 64-query items by waves 0,1 over the whole key range (same arithmetic as a full item).
// ------------------------------------------------------------------------------------------------
template <bool GATE>
__global__ __launch_bounds__(256, 2) void k_attn_f32(const float* __restrict__ qkvg, int ld, float* __restrict__ out, int ldo,
                                                     const int* __restrict__ cu, const int* __restrict__ qblocks, int d_model, int gqa,
                                                     int rep, float c_exp /* scale * log2(e) */, float* __restrict__ lse_out) {
  __shared__ __attribute__((aligned(16))) uint4 kl[KB * 16];   // [key][16 chunks of 4 floats], chunk c at c ^ (key & 15)
  __shared__ __attribute__((aligned(16))) float vl[KB * 64];   // [key][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int seq = qblocks[4 * blockIdx.x], q0 = qblocks[4 * blockIdx.x + 1], head = qblocks[4 * blockIdx.x + 2];
  const int mode = qblocks[4 * blockIdx.x + 3];
  if (seq < 0) return;
  const int s0 = cu[seq], S = cu[seq + 1] - s0;
  const int q_lim = (mode && q0 + 64 < S) ? q0 + 64 : S;      // half item: 64 query rows
  const bool wave_live = !(mode && wave >= 2);                 // its waves 2,3 only help staging
  const int kvh = head / rep;
  const float* qbase = qkvg + (size_t)s0 * ld + head * 64;
  const float* gbase = qkvg + (size_t)s0 * ld + d_model + head * 64;
  const float* kbase = qkvg + (size_t)s0 * ld + 2 * d_model + kvh * 64;
  const float* vbase = kbase + gqa;

  // Q fragments: lane holds Q[query r][32h + 0..31]
  const int qrow = q0 + wave * 32 + r;
  const int qrc = qrow < S ? qrow : S - 1;
  f32x4 qf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const f32x4*>(qbase + (size_t)qrc * ld + h * 32 + c * 4);

  // staging: a K / V tile is 64 keys x 16 chunks; thread t takes chunk t & 15 of keys (t >> 4) + 16 i
  const int skey = tid >> 4, sch = tid & 15;
  uint4 sk0, sk1, sk2, sk3, sv0, sv1, sv2, sv3;
#define A32_GLOAD(kt_)                                                                     \
  do {                                                                                     \
    int k0__ = (kt_) * KB + skey, k1__ = k0__ + 16, k2__ = k0__ + 32, k3__ = k0__ + 48;    \
    k0__ = k0__ < S ? k0__ : S - 1; k1__ = k1__ < S ? k1__ : S - 1;                        \
    k2__ = k2__ < S ? k2__ : S - 1; k3__ = k3__ < S ? k3__ : S - 1;                        \
    sk0 = *reinterpret_cast<const uint4*>(kbase + (size_t)k0__ * ld + sch * 4);            \
    sk1 = *reinterpret_cast<const uint4*>(kbase + (size_t)k1__ * ld + sch * 4);            \
    sk2 = *reinterpret_cast<const uint4*>(kbase + (size_t)k2__ * ld + sch * 4);            \
    sk3 = *reinterpret_cast<const uint4*>(kbase + (size_t)k3__ * ld + sch * 4);            \
    sv0 = *reinterpret_cast<const uint4*>(vbase + (size_t)k0__ * ld + sch * 4);            \
    sv1 = *reinterpret_cast<const uint4*>(vbase + (size_t)k1__ * ld + sch * 4);            \
    sv2 = *reinterpret_cast<const uint4*>(vbase + (size_t)k2__ * ld + sch * 4);            \
    sv3 = *reinterpret_cast<const uint4*>(vbase + (size_t)k3__ * ld + sch * 4);            \
  } while (0)
  // (skey + 16 i) & 15 == skey: one swizzled chunk position for the four rows
  const int kli = skey * 16 + (sch ^ skey);
  uint4* vl4 = reinterpret_cast<uint4*>(vl);
#define A32_LSTORE()                                                                       \
  do {                                                                                     \
    kl[kli] = sk0; kl[kli + 256] = sk1; kl[kli + 512] = sk2; kl[kli + 768] = sk3;          \
    vl4[tid] = sv0; vl4[tid + 256] = sv1; vl4[tid + 512] = sv2; vl4[tid + 768] = sv3;      \
  } while (0)

  f32x16 o_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + KB - 1) / KB;
  A32_GLOAD(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();            // every wave is done with the previous tile
    A32_LSTORE();
    __syncthreads();
    if (kt + 1 < nkt) A32_GLOAD(kt + 1);   // in flight behind this tile's MFMAs
    if (wave_live) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {        // two 32-key sub-tiles
        if (kt * KB + t * 32 >= S) break;  // wave-uniform
        // ---- S^T = K Q^T ----
        f32x16 sc = zero16;
        const int key = t * 32 + r;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const f32x4 kf = __builtin_bit_cast(f32x4, kl[key * 16 + ((h * 8 + c) ^ (key & 15))]);
#pragma unroll
          for (int j = 0; j < 4; ++j) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[c][j], sc, 0, 0, 0);
        }
        if (kt * KB + t * 32 + 32 > S) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int kk = kt * KB + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (kk >= S) sc[e] = -INFINITY;
          }
        }
        // ---- online softmax ----
        float mx = sc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sc[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float mc = m_new * c_exp;
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          sc[e] = __builtin_amdgcn_exp2f(fmaf(sc[e], c_exp, -mc));
          psum += sc[e];
        }
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
          l_run *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
          m_run = m_new;
        }
        l_run += psum;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int vk = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v0 = vl[vk * 64 + r], v1 = vl[vk * 64 + 32 + r];
          o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sc[e], o_acc[0], 0, 0, 0);
          o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sc[e], o_acc[1], 0, 0, 0);
        }
      }
    }
  }
#undef A32_GLOAD
#undef A32_LSTORE
  if (!wave_live) return;

  // ---- normalise, gate, store: lane holds O[query r][32dt + 8g + 4h + 0..3] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.0f / l_tot;
  if (lse_out && qrow < q_lim && h == 0)
    lse_out[(size_t)(s0 + qrow) * (d_model >> 6) + head] = m_run * (c_exp * 0.69314718055994530942f) + logf(l_tot);
  if (qrow < q_lim) {
    float* orow = out + (size_t)(s0 + qrow) * ldo + head * 64;
    const float* grow = gbase + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        f32x4 v = {o_acc[dt][4 * g] * inv_l, o_acc[dt][4 * g + 1] * inv_l, o_acc[dt][4 * g + 2] * inv_l, o_acc[dt][4 * g + 3] * inv_l};
        if (GATE) {
          const f32x4 gt = *reinterpret_cast<const f32x4*>(grow + d0);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= 1.0f / (1.0f + expf(-gt[e]));
        }
        *reinterpret_cast<f32x4*>(orow + d0) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 ("three-pass") attention for fp32 towers (flag TTV_ATTN_SPLIT3; round 4).  fp32 in, fp32 out, fp32 softmax - as k_attn_f32 -
// but both products run on the bf16 matrix pipe with every operand split into hi + lo (hi = bf16(x), lo = bf16(x - hi)):
//     S^T = Kh Qh^T + Kh Ql^T + Kl Qh^T        O^T += Vh^T Ph^T + Vh^T Pl^T + Vl^T Ph^T        (fp32 accumulation)
// ~2^-17 relative per product instead of bf16's 2^-9: with the split GEMMs (k_gemm_f32<.., SPLIT>) the encoder keeps every token index
// of the reference's fp32 run on the benchmark fixture at about a third of the exact-fp32 kernels' time (tests/probes/split_bf16_probe.py,
// bench.py `exact_index`).  Structure: k_attn_bf16's tiles and fragment reads (K rows XOR-swizzled for ds_read_b128, V^T through
// ds_read_b64_tr_b16) on FOUR bf16 tiles - K hi / lo, V hi / lo - filled by register staging: a thread loads 16 bytes of fp32, splits
// them and stores the two 8-byte halves (the next tile's loads are in flight behind the MFMAs).  Exact running maximum, exact row sums
// of the fp32 p (only the operands of the two products are split).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split4_f32(const uint4 v, uint2& hi, uint2& lo) {
  const float x0 = __uint_as_float(v.x), x1 = __uint_as_float(v.y), x2 = __uint_as_float(v.z), x3 = __uint_as_float(v.w);
  const bf16_t h0 = (bf16_t)x0, h1 = (bf16_t)x1, h2 = (bf16_t)x2, h3 = (bf16_t)x3;
  const bf16x4 hv = {h0, h1, h2, h3};
  const bf16x4 lv = {(bf16_t)(x0 - (float)h0), (bf16_t)(x1 - (float)h1), (bf16_t)(x2 - (float)h2), (bf16_t)(x3 - (float)h3)};
  hi = __builtin_bit_cast(uint2, hv);
  lo = __builtin_bit_cast(uint2, lv);
}

// IMG (round 4): q, k and v arrive as the to_qkv epilogue's planar images (per 8 features hi0..7 | lo0..7; GemmArgs.y_image = 2): the Q
// fragments are two 16-byte loads each, and the four bf16 tiles of a stage are filled by LDS-DMA (a 16-byte chunk of a tile row IS a
// 16-byte chunk of the image; swizzle on the source side) - no staging registers, no ds_write, no split arithmetic for K and V.
template <bool GATE, bool IMG = false>
__global__ __launch_bounds__(256, 2) void k_attn_split3(const float* __restrict__ qkvg, int ld, float* __restrict__ out, int ldo,
                                                        const int* __restrict__ cu, const int* __restrict__ qblocks, int d_model, int gqa,
                                                        int rep, float c_exp /* scale * log2(e) */, int out_image) {
  // out_image: the output is written as out_proj's split image (hi0..3 | lo0..3 per four consecutive features) instead of fp32
  __shared__ __attribute__((aligned(16))) uint4 tiles[2][4][KB * 8];   // two stages of (K hi, K lo, V hi, V lo), 64 keys x 128 bytes each
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int seq = qblocks[4 * blockIdx.x], q0 = qblocks[4 * blockIdx.x + 1], head = qblocks[4 * blockIdx.x + 2];
  const int mode = qblocks[4 * blockIdx.x + 3];
  if (seq < 0) return;
  const int s0 = cu[seq], S = cu[seq + 1] - s0;
  const int q_lim = (mode && q0 + 64 < S) ? q0 + 64 : S;      // half item: 64 query rows, waves 0 and 1 over the whole key range
  const bool wave_live = !(mode && wave >= 2);                 // its waves 2, 3 only help staging
  const int kvh = head / rep;
  const float* qbase = qkvg + (size_t)s0 * ld + head * 64;
  const float* gbase = qkvg + (size_t)s0 * ld + d_model + head * 64;
  const float* kbase = qkvg + (size_t)s0 * ld + 2 * d_model + kvh * 64;
  const float* vbase = kbase + gqa;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query r][16 ks + 8 h + 0..7], split once
  const int qrow = q0 + wave * 32 + r;
  const int qrc = qrow < S ? qrow : S - 1;
  bf16x8 qh[4], ql[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint4 a = *reinterpret_cast<const uint4*>(qbase + (size_t)qrc * ld + ks * 16 + h * 8);
    const uint4 b = *reinterpret_cast<const uint4*>(qbase + (size_t)qrc * ld + ks * 16 + h * 8 + 4);
    if (IMG) {                    // the group of 8 features IS (hi0..7 | lo0..7)
      qh[ks] = __builtin_bit_cast(bf16x8, a);
      ql[ks] = __builtin_bit_cast(bf16x8, b);
    } else {
      uint2 ah, al, bh, bl;
      split4_f32(a, ah, al);
      split4_f32(b, bh, bl);
      qh[ks] = __builtin_bit_cast(bf16x8, make_uint4(ah.x, ah.y, bh.x, bh.y));
      ql[ks] = __builtin_bit_cast(bf16x8, make_uint4(al.x, al.y, bl.x, bl.y));
    }
  }

  // staging: a tile is 64 keys x 16 float chunks; thread t takes float chunk t & 15 (dims 4 sch .. 4 sch + 3) of keys (t >> 4) + 16 i
  const int skey = tid >> 4, sch = tid & 15;
  uint4 sk[4], sv[4];
#define S3_GLOAD(kt_)                                                                      \
  do {                                                                                     \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                  \
      int k__ = (kt_) * KB + skey + 16 * i__;                                              \
      k__ = k__ < S ? k__ : S - 1;                                                         \
      sk[i__] = *reinterpret_cast<const uint4*>(kbase + (size_t)k__ * ld + sch * 4);       \
      sv[i__] = *reinterpret_cast<const uint4*>(vbase + (size_t)k__ * ld + sch * 4);       \
    }                                                                                      \
  } while (0)
  // bf16 tile row = 128 bytes = 8 chunks of 16 bytes; float chunk sch is the 8-byte half sch & 1 of chunk sch >> 1, stored at the
  // swizzled chunk position k_attn_bf16's fragment reads expect: K: c ^ ((row >> 1) & 7), V: c ^ (((row >> 1) & 1) << 2).
  // Rows skey + 16 i: (row >> 1) & 7 and (row >> 1) & 1 do not depend on i.
  char* const tile0 = reinterpret_cast<char*>(&tiles[0][0][0]);
  constexpr int STAGE = 4 * KB * 128, PLANE = KB * 128;
  const int kst = skey * 128 + (((sch >> 1) ^ ((skey >> 1) & 7)) << 4) + (sch & 1) * 8;
  const int vst = skey * 128 + (((sch >> 1) ^ (((skey >> 1) & 1) << 2)) << 4) + (sch & 1) * 8;
#define S3_LSTORE(stage_)                                                                  \
  do {                                                                                     \
    char* const sb__ = tile0 + (stage_) * STAGE;                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                  \
      uint2 hi__, lo__;                                                                    \
      split4_f32(sk[i__], hi__, lo__);                                                     \
      *reinterpret_cast<uint2*>(sb__ + kst + i__ * 2048) = hi__;                           \
      *reinterpret_cast<uint2*>(sb__ + PLANE + kst + i__ * 2048) = lo__;                   \
      split4_f32(sv[i__], hi__, lo__);                                                     \
      *reinterpret_cast<uint2*>(sb__ + 2 * PLANE + vst + i__ * 2048) = hi__;               \
      *reinterpret_cast<uint2*>(sb__ + 3 * PLANE + vst + i__ * 2048) = lo__;               \
    }                                                                                      \
  } while (0)

  // fragment offsets exactly as in k_attn_bf16
  const int ksw = (r >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((ks * 2 + h) ^ ksw) << 4);
  const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
  const int vsw = (tq >> 1) & 1;
  const int vlane = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
  const int voff_d0 = vlane + (vsw ? 64 : 0), voff_d1 = vlane + (vsw ? 0 : 64);

  f32x16 o_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + KB - 1) / KB;
  // IMG: LDS-DMA staging.  Plane pl (K hi, K lo, V hi, V lo) of a stage is 64 rows x 8 chunks; one instruction fills 8 rows (lane >> 3 =
  // row, lane & 7 = chunk position), wave w rows 16 w .. 16 w + 15 of every plane: 8 instructions per wave and tile.  Position p of a K
  // row holds logical chunk p ^ ((row >> 1) & 7), of a V row p ^ (((row >> 1) & 1) << 2); logical chunk c of plane hi / lo is image bytes
  // 32 c / 32 c + 16 of the row's 64 features.
  const uint32_t tiles_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&tiles[0][0][0];
  const int drow = lane >> 3, dpos = lane & 7;
#define S3_DMA(voff_, base_, dst_)                                                                               \
  do {                                                                                                           \
    unsigned keep__;                                                                                             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep__) : "v"(voff_), "s"(base_), "s"(dst_) : "memory");                                 \
  } while (0)
#define S3_DMA_TILE(kt_, stage_)                                                                                 \
  do {                                                                                                           \
    _Pragma("unroll") for (int i__ = 0; i__ < 2; ++i__) {                                                        \
      const int row__ = wave * 16 + i__ * 8 + drow;                                                              \
      int key__ = (kt_) * KB + row__;                                                                            \
      key__ = key__ < S ? key__ : S - 1;                                                                         \
      const uint32_t rb__ = (uint32_t)key__ * (uint32_t)ld * 4u;                                                 \
      const uint32_t kc__ = (uint32_t)((dpos ^ ((row__ >> 1) & 7)) * 32);                                        \
      const uint32_t vc__ = (uint32_t)((dpos ^ (((row__ >> 1) & 1) << 2)) * 32);                                 \
      const uint32_t dst__ = tiles_lds + (stage_) * STAGE + (wave * 16 + i__ * 8) * 128;                         \
      S3_DMA(rb__ + kc__, kbase, dst__);                                                                         \
      S3_DMA(rb__ + kc__ + 16u, kbase, dst__ + PLANE);                                                           \
      S3_DMA(rb__ + vc__, vbase, dst__ + 2 * PLANE);                                                             \
      S3_DMA(rb__ + vc__ + 16u, vbase, dst__ + 3 * PLANE);                                                       \
    }                                                                                                            \
  } while (0)
  // two LDS stages, one barrier per tile: while a wave computes on stage kt & 1 it has already written tile kt + 1 into the other stage
  // (from the registers loaded one tile earlier) and has tile kt + 2's global loads in flight
  if (IMG) {
    S3_DMA_TILE(0, 0);
  } else {
    S3_GLOAD(0);
    S3_LSTORE(0);
    if (nkt > 1) S3_GLOAD(1);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    if (IMG) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of tile kt has landed
    __syncthreads();            // stage kt & 1 is complete; every wave is done computing on the other stage
    if (IMG) {
      if (kt + 1 < nkt) S3_DMA_TILE(kt + 1, (kt + 1) & 1);
    } else if (kt + 1 < nkt) {
      S3_LSTORE((kt + 1) & 1);
      if (kt + 2 < nkt) S3_GLOAD(kt + 2);
    }
    if (!wave_live) continue;
    const char* const tk_h = tile0 + (kt & 1) * STAGE;
    const char* const tk_l = tk_h + PLANE;
    const char* const tv_h = tk_h + 2 * PLANE;
    const char* const tv_l = tk_h + 3 * PLANE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {          // two 32-key sub-tiles
      if (kt * KB + t * 32 >= S) break;    // wave-uniform
      // ---- S^T = K Q^T, three passes (cross terms first) ----
      f32x16 sc = zero16;
      {
        bf16x8 kfh[4], kfl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          kfh[ks] = *reinterpret_cast<const bf16x8*>(tk_h + koff[ks] + t * 4096);
          kfl[ks] = *reinterpret_cast<const bf16x8*>(tk_l + koff[ks] + t * 4096);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl[ks], qh[ks], sc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh[ks], ql[ks], sc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh[ks], qh[ks], sc, 0, 0, 0);
      }
      if (kt * KB + t * 32 + 32 > S) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kk = kt * KB + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (kk >= S) sc[e] = -INFINITY;
        }
      }
      // ---- online softmax (fp32, exact running maximum) ----
      float mx = sc[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sc[e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float mc = m_new * c_exp;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        sc[e] = __builtin_amdgcn_exp2f(fmaf(sc[e], c_exp, -mc));
        psum += sc[e];
      }
      if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
        m_run = m_new;
      }
      l_run += psum;
      // ---- P split into bf16 B-operand fragments: k-step sp = registers 8 sp .. 8 sp + 7 ----
      bf16x8 ph[2], pl[2];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16_t hv = (bf16_t)sc[8 * sp + j];
          ph[sp][j] = hv;
          pl[sp][j] = (bf16_t)(sc[8 * sp + j] - (float)hv);
        }
      // ---- O^T += V^T P^T, three passes ----
#define S3_VFRAG(base_, dt_, sp_)                                                                                  \
  ({                                                                                                               \
    const char* vb__ = (base_) + ((dt_) == 0 ? voff_d0 : voff_d1) + t * 4096 + (sp_) * 2048;                       \
    const bf16x4 lo__ = lds_read_tr16(vb__), hi__ = lds_read_tr16(vb__ + 1024);                                    \
    (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                              \
  })
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 vh0 = S3_VFRAG(tv_h, dt, 0), vh1 = S3_VFRAG(tv_h, dt, 1);
        const bf16x8 vl0 = S3_VFRAG(tv_l, dt, 0), vl1 = S3_VFRAG(tv_l, dt, 1);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0, ph[0], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1, ph[1], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0, pl[0], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1, pl[1], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0, ph[0], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1, ph[1], o_acc[dt], 0, 0, 0);
      }
#undef S3_VFRAG
    }
  }
#undef S3_GLOAD
#undef S3_LSTORE
#undef S3_DMA_TILE
#undef S3_DMA
  if (!wave_live) return;

  // ---- normalise, gate, store (fp32, exact division and sigmoid as k_attn_f32): lane holds O[query r][32 dt + 8 g + 4 h + 0..3] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.0f / l_tot;
  if (qrow < q_lim) {
    float* orow = out + (size_t)(s0 + qrow) * ldo + head * 64;
    const float* grow = gbase + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        f32x4 v = {o_acc[dt][4 * g] * inv_l, o_acc[dt][4 * g + 1] * inv_l, o_acc[dt][4 * g + 2] * inv_l, o_acc[dt][4 * g + 3] * inv_l};
        if (GATE) {
          const f32x4 gt = *reinterpret_cast<const f32x4*>(grow + d0);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= 1.0f / (1.0f + expf(-gt[e]));
        }
        if (out_image) {
          uint2 hi, lo;
          split4_f32(__builtin_bit_cast(uint4, v), hi, lo);
          *reinterpret_cast<uint4*>(orow + d0) = make_uint4(hi.x, hi.y, lo.x, lo.y);
        } else {
          *reinterpret_cast<f32x4*>(orow + d0) = v;
        }
      }
  }
}

// Attention whose gated output leaves only as the block-scaled e4m3 operand of out_proj (k_attn_bf16<.., MXO>): bf16 inputs with
// pre-scaled q, unpaired table, sigmoid gate.  out_q [L, d_model] bytes, out_mx [L, ld_mx] E8M0 scales (ttvk_mx_scale_ld(d_model)).
int ttvk_attention_mxout(const void* qkvg, int ld, void* out_q, void* out_mx, int ld_mx, const int* cu_seqlens, const int* qblocks,
                         int n_qblocks, int q_heads, int kv_heads, hipStream_t s) {
  if (n_qblocks == 0) return TTV_OK;
  TTV_CHECK_ARG(kv_heads > 0 && q_heads % kv_heads == 0, "attention_mxout: q_heads %% kv_heads");
  const int d_model = q_heads * 64, gqa = kv_heads * 64, rep = q_heads / kv_heads;
  TTV_CHECK_ARG(ld >= 2 * d_model + 2 * gqa && ld % 8 == 0 && d_model % 128 == 0 && ld_mx == (int)(4 * ((d_model / 128 + 3) / 4 * 4)),
                "attention_mxout: bad leading dims (d_model %% 128, ld_mx = 4 * round_up(d_model / 128, 4))");
  TTV_CHECK_ARG(qkvg && out_q && out_mx && (uintptr_t)qkvg % 16 == 0 && (uintptr_t)out_q % 8 == 0, "attention_mxout: null / unaligned pointers");
  static const float defer_thr = getenv("TTV_ATTN_THR") ? (float)atof(getenv("TTV_ATTN_THR")) : ATTN_DEFER_THR;
  TtvProfScope prof(TTV_KC_ATTENTION, s);
  hipLaunchKernelGGL((k_attn_bf16<true, 1, true, false, true>), dim3(n_qblocks), dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)nullptr, ld_mx,
                     cu_seqlens, qblocks, n_qblocks, d_model, gqa, rep, 1.0f, (float*)out_mx, (bf16_t*)out_q, defer_thr, g_ttv_stamps);
  TTV_CHECK_LAUNCH("attention_mxout");
  return TTV_OK;
}

// flags: bit 0 (TTV_ATTN_GATE) multiply by sigmoid(gate); bit 1 (TTV_ATTN_PAIRED) the table is paired (see k_attn_bf16, NE = 2)
// out_raw (bf16 only, with TTV_ATTN_GATE; leading dimension ldo): additionally receives the UNGATED attention output
int ttvk_attention(const void* qkvg, int ld, void* out, int ldo, const int* cu_seqlens, const int* qblocks, int n_qblocks,
                   int q_heads, int kv_heads, int head_dim, int flags, int dtype, hipStream_t s, float* lse_out, void* out_raw) {
  const int gate_mul = flags & 1, paired = (flags >> 1) & 1, prescaled = (flags >> 2) & 1;
  TTV_CHECK_ARG(!prescaled || dtype == TTV_BF16, "attention: pre-scaled q is a bf16 path option");
  TTV_CHECK_ARG(!out_raw || (dtype == TTV_BF16 && gate_mul), "attention: the second (ungated) output is a bf16 + gate option");
  if (n_qblocks == 0) return TTV_OK;
  TTV_CHECK_ARG(head_dim == 64, "attention: head_dim %d unsupported (the reference fixes 64, utils.py:8)", head_dim);
  TTV_CHECK_ARG(kv_heads > 0 && q_heads % kv_heads == 0, "attention: q_heads %% kv_heads");
  const int d_model = q_heads * 64, gqa = kv_heads * 64, rep = q_heads / kv_heads;
  TTV_CHECK_ARG(ld >= 2 * d_model + 2 * gqa && ld % 8 == 0 && ldo % 4 == 0, "attention: bad leading dims");
  TTV_CHECK_ARG((uintptr_t)qkvg % 16 == 0 && (uintptr_t)out % (dtype == TTV_F32 ? 16 : 8) == 0, "attention: unaligned pointers");
  dim3 grid(n_qblocks);   // one entry per (sequence, 128-query block, q-head), XCD-interleaved by the host
  const float scale = 0.125f;  // 64^-0.5
  TtvProfScope prof(TTV_KC_ATTENTION, s);
  if (dtype == TTV_BF16) {
    const float c_exp = scale * 1.44269504088896340736f;
    // pre-scaled q: the exponent factor is 1; the accumulator-carried maximum (PRE) is built for the 4-wave kernel only - the paired
    // kernel is held to 128 VGPRs and would spill its start vector - so paired launches run the generic softmax with factor 1
    const float c_eff = prescaled ? 1.0f : c_exp;
    static const float defer_thr = getenv("TTV_ATTN_THR") ? (float)atof(getenv("TTV_ATTN_THR")) : ATTN_DEFER_THR;   // diagnostics: 0 = exact running maximum
#define ATTN_LAUNCH(G_, NE_, P_, T_, grid_, threads_)                                                                                   \
  hipLaunchKernelGGL((k_attn_bf16<G_, NE_, (P_) && (NE_) == 1, T_>), grid_, dim3(threads_), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, \
                     cu_seqlens, qblocks, n_qblocks, d_model, gqa, rep, c_eff, lse_out, (bf16_t*)out_raw, defer_thr, g_ttv_stamps)
#define ATTN_PICK(NE_, grid_, threads_)                                                  \
  do {                                                                                   \
    if (tape) { if (gate_mul) ATTN_LAUNCH(true, NE_, false, true, grid_, threads_); else ATTN_LAUNCH(false, NE_, false, true, grid_, threads_); }  \
    else if (gate_mul) { if (prescaled) ATTN_LAUNCH(true, NE_, true, false, grid_, threads_); else ATTN_LAUNCH(true, NE_, false, false, grid_, threads_); }   \
    else { if (prescaled) ATTN_LAUNCH(false, NE_, true, false, grid_, threads_); else ATTN_LAUNCH(false, NE_, false, false, grid_, threads_); }          \
  } while (0)
    const bool tape = lse_out != nullptr || out_raw != nullptr;
    TTV_CHECK_ARG(!tape || !prescaled, "attention: the training-tape outputs need unscaled q");
    // tables of full items only with pre-scaled q, no tape: the software-pipelined kernel on request (flag TTV_ATTN_PIPE; slower
    // than the plain loop, see its header)
    // round 5: the in-wave software pipeline of ttv_attn_swp.hip is the default for such tables (TTV_ATTN_SWP=0: k_attn_bf16, A/B)
    static const bool swp_env = !(getenv("TTV_ATTN_SWP") && getenv("TTV_ATTN_SWP")[0] == '0');
    if (swp_env && !(flags & TTV_ATTN_PIPE) && (flags & TTV_ATTN_ALLFULL) && prescaled && !paired && !tape && !(g_ttv_debug & 1048576)) {
      return ttvk_attention_swp(qkvg, ld, out, ldo, cu_seqlens, qblocks, n_qblocks, q_heads, kv_heads, gate_mul, s);
    }
    if ((flags & TTV_ATTN_PIPE) && (flags & TTV_ATTN_ALLFULL) && prescaled && !paired && !tape) {
      // the reference moves when a lane's 16-key sum of p exceeds 2^pipe_thr: bf16 P and fp32 sums have the range for it, and
      // with 40 the rare branch is rare for any score distribution (8, k_attn_bf16's value: every few tiles at a spread of 6)
      static const float pipe_thr = getenv("TTV_ATTN_PIPE_THR") ? (float)atof(getenv("TTV_ATTN_PIPE_THR")) : ATTN_PIPE_THR;
      if (gate_mul) hipLaunchKernelGGL((k_attn_pipe<true>), grid, dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, pipe_thr, g_ttv_stamps);
      else hipLaunchKernelGGL((k_attn_pipe<false>), grid, dim3(256), 0, s, (const bf16_t*)qkvg, ld, (bf16_t*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, pipe_thr, g_ttv_stamps);
    } else if (paired) {
      // rows of 8 list slots; a block takes two consecutive rows of one slot
      const int rows = ttv_cdiv(n_qblocks, 8), pairs = ttv_cdiv(rows, 2);
      dim3 g2(pairs * 8);
      ATTN_PICK(2, g2, 512);
    } else {
      ATTN_PICK(1, grid, 256);
    }
#undef ATTN_PICK
#undef ATTN_LAUNCH
  } else if (dtype == TTV_F32) {
    const float c_exp = scale * 1.44269504088896340736f;
    if (flags & TTV_ATTN_SPLIT3) {
      TTV_CHECK_ARG(!lse_out, "attention: the split-bf16 kernel is an inference path (no tape outputs)");
      const int img = (flags & TTV_ATTN_SPLIT_OUT) ? 1 : 0;
      if (flags & TTV_ATTN_SPLIT_IN) {
        if (gate_mul) hipLaunchKernelGGL((k_attn_split3<true, true>), grid, dim3(256), 0, s, (const float*)qkvg, ld, (float*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, c_exp, img);
        else hipLaunchKernelGGL((k_attn_split3<false, true>), grid, dim3(256), 0, s, (const float*)qkvg, ld, (float*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, c_exp, img);
      } else if (gate_mul) hipLaunchKernelGGL((k_attn_split3<true>), grid, dim3(256), 0, s, (const float*)qkvg, ld, (float*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, c_exp, img);
      else hipLaunchKernelGGL((k_attn_split3<false>), grid, dim3(256), 0, s, (const float*)qkvg, ld, (float*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, c_exp, img);
    } else if (gate_mul)
      hipLaunchKernelGGL((k_attn_f32<true>), grid, dim3(256), 0, s, (const float*)qkvg, ld, (float*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, c_exp, lse_out);
    else
      hipLaunchKernelGGL((k_attn_f32<false>), grid, dim3(256), 0, s, (const float*)qkvg, ld, (float*)out, ldo, cu_seqlens, qblocks, d_model, gqa, rep, c_exp, lse_out);
  } else {
    ttv_set_error("attention: bad dtype");
    return TTV_ERR_INVALID;
  }
  TTV_CHECK_LAUNCH("attention");
  return TTV_OK;
}

// Fused GEGLU feed-forward sub-layer of a width-256 tower, one kernel (reference model/base/transformer.py:47-56 + the
// residual / KEEL step at :130 and :144-145):
//
//     h     = gelu_erf(x_n W12g^T) * (x_n W12x^T)          x_n = RMSNorm(x) * norm.weight   (folded, see below)
//     y     = alpha * x + h W3^T
//     x_new = KEEL ? RMSNorm(y) * post_gain : y
//
// Why one kernel: unfused, this sub-layer moves x (19 MB) -> h (52 MB write + 52 MB read) -> x at the benchmark shape and
// pays two launches; fused it reads x once, writes x_new once (38 MB) and h never leaves the CU.
//
// Structure (bf16, K = d = 256, I % 32 == 0): see k_mlp256 below.
//   * the pre-norm is folded (gain pre-multiplied into W12's columns on the host, rstd from the register-resident row).
//   * the hidden dimension is walked in panels of 32 (x, gate) pairs.  Per panel the host-packed image (ttv_mlp_pack:
//     a 64-row W12 panel already XOR-swizzled, 32 KiB, + the matching 256 x 32 W3 slice chunk-major, 16 KiB) is copied
//     L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB contiguous per wave-instruction), double-buffered.
//   * GEGLU in registers: with mfma_16x16x32's C layout a lane holds, for its token, features {4kq..4kq+3} of m-tile 0
//     and {16+4kq..16+4kq+3} of m-tile 1: exactly 8 values = one B fragment of the second product, provided the k order
//     of that product is  k(kq, j) = j<4 ? 4kq+j : 16+4kq+(j-4).  The W3 slice is packed in that order, so the hidden
//     tile goes producer lane -> LDS -> the same lane of the consumer wave as one 16-byte vector, no shuffles.
//   * epilogue: residual reload (L2-hot), alpha*x + acc, row statistics, gain, lane-pair exchange, 16-byte stores.
#include "ttv_common.h"
#include "ttv_kernels.h"

#define MLP_W12_CHUNKS 2048   // uint4 per W12 panel image (64 rows x 32 chunks)
#define MLP_W3_CHUNKS 1024    // uint4 per W3 slice image (4 k-chunks x 256 rows)
#define MLP_PANEL_CHUNKS (MLP_W12_CHUNKS + MLP_W3_CHUNKS)
#define MLP_WO_CHUNKS (4 * MLP_W12_CHUNKS)   // out_proj: 4 panel images of 64 rows, appended after the I/32 feed-forward panels

struct MlpDev {
  const bf16_t* x; int ldx;
  const uint4* pack;   // [I/32][MLP_PANEL_CHUNKS] panel images, then [4][MLP_W12_CHUNKS] out_proj images (ttv_mlp_pack)
  // optional fused front (attention out_proj + residual/KEEL, transformer.py:104,141-142): x <- [RMSNorm](fa*x + ao Wo^T)[*fg]
  const bf16_t* ao; int ldao;
  const float* front_gain;
  float front_alpha;
  // optional fused back (the NEXT layer's pre_ln + to_qkv + rotary, transformer.py:86-98): qkv <- rope(RMSNorm(y) @ Wqkv'^T),
  // Wqkv' = to_qkv * pre_ln gain, packed as nqp 64-row panel images after the out_proj images
  bf16_t* qkv; int ldq;
  const float* rope_cs;
  int nqp, rope_q_end, rope_k_begin, rope_k_end;
  int I;
  bf16_t* y; int ldy;
  const float* post_gain;
  float alpha, eps;
  int M, n_tiles, debug;
  long long* stamps;   // diagnostics (ttv_debug_stamps): [2 roles][64] s_memtime values of block 0
};

// ---- weight packing: builds the per-panel LDS images once per weight version -------------------------------------
__global__ void k_mlp_pack(const bf16_t* __restrict__ w12f, const bf16_t* __restrict__ w3, const bf16_t* __restrict__ wo,
                           const bf16_t* __restrict__ wq, int nq_rows, int I, uint4* __restrict__ out) {
  const int np = I / 32;
  const int nqp = (nq_rows + 63) / 64;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= np * MLP_PANEL_CHUNKS + MLP_WO_CHUNKS + nqp * MLP_W12_CHUNKS) return;
  if (o >= np * MLP_PANEL_CHUNKS + MLP_WO_CHUNKS) {
    // next layer's to_qkv (pre-norm gain folded) as 64-row panel images, k order of the C-layout B fragments
    const int c = o - np * MLP_PANEL_CHUNKS - MLP_WO_CHUNKS, qp = c / MLP_W12_CHUNKS, cc = c - qp * MLP_W12_CHUNKS;
    const int r = cc >> 5, cp = cc & 31;
    const int ch = (cp & 16) | ((cp & 15) ^ (r & 15));
    int srow = qp * 64 + r;
    srow = srow < nq_rows ? srow : nq_rows - 1;
    const bf16_t* src = wq + (size_t)srow * 256 + (ch >> 2) * 32 + (ch & 3) * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
    out[o] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return;
  }
  if (o >= np * MLP_PANEL_CHUNKS) {
    // out_proj images: 4 panels of 64 rows, natural k order, same XOR swizzle
    const int c = o - np * MLP_PANEL_CHUNKS, wp = c / MLP_W12_CHUNKS, cc = c - wp * MLP_W12_CHUNKS;
    const int r = cc >> 5, cp = cc & 31;
    const int ch = (cp & 16) | ((cp & 15) ^ (r & 15));
    out[o] = wo ? *reinterpret_cast<const uint4*>(wo + (size_t)(wp * 64 + r) * 256 + ch * 8) : make_uint4(0, 0, 0, 0);
    return;
  }
  const int pn = o / MLP_PANEL_CHUNKS, c = o - pn * MLP_PANEL_CHUNKS;
  if (c < MLP_W12_CHUNKS) {
    const int r = c >> 5, cp = c & 31;
    const int ch = (cp & 16) | ((cp & 15) ^ (r & 15));   // LDS chunk cp of row r holds source chunk ch = 4 s8 + kq
    const int srow = r < 32 ? pn * 32 + r : I + pn * 32 + (r - 32);
    // k order inside every 32-column step follows the B fragments the kernel builds from C-layout registers:
    // lane group kq holds columns {4kq..4kq+3} and {16+4kq..16+4kq+3} of the step
    const bf16_t* src = w12f + (size_t)srow * 256 + (ch >> 2) * 32 + (ch & 3) * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
    out[o] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  } else {
    const int c3 = c - MLP_W12_CHUNKS, kq = c3 >> 8, row = c3 & 255;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w3[(size_t)row * I + pn * 32 + (j < 4 ? 4 * kq + j : 16 + 4 * kq + (j - 4))];
    out[o] = __builtin_bit_cast(uint4, v);
  }
}

// Block of 8 waves (two per SIMD, 256 VGPRs each, no AGPRs).  Waves w and w+4 share a SIMD and a group of NT*16 tokens;
// neither can hold both the x rows (NT x 8 x 16 B of B fragments) and all y accumulators (16 m-tiles x NT x 4) of the
// group, so the token tiles are split between them in complementary roles:
//     "P1" of a tile: x rows register-resident; per panel 4 m-tiles x 8 k-steps MFMAs -> GEGLU in registers -> the 8 hidden
//                     values a lane ends up with are exactly one B fragment of the second product -> LDS, one 16-byte vector.
//     "P2" of a tile: the 16 m-tile accumulators of y^T (all 256 features -> the KEEL RMSNorm is wave-local); per panel
//                     16 MFMAs on the fragment its partner produced ONE iteration earlier; residual / norm / store epilogue.
//   NT = 3: wave w   = P1 of tiles {0,1} + P2 of tile {2};   wave w+4 = P1 of tile {2} + P2 of tiles {0,1}
//   NT = 2: wave w   = P1 of tile {0}    + P2 of tile {1};   wave w+4 = P1 of tile {1} + P2 of tile {0}
// Both waves carry MFMA and VALU (GEGLU) work in different proportions and drift against each other, so the SIMD's matrix
// pipe runs one wave's MFMAs under the other's GEGLU without software pipelining.  Program order inside an iteration:
// P1 MFMAs, P2 MFMAs (previous panel), GEGLU - the P2 MFMAs cover the latency of the P1 results the GEGLU needs.
// One barrier per panel; weight images arrive by LDS-DMA one iteration ahead of their use.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 cvt4(bf16x4 b) { return (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]}; }
#define MLP_LD12(img_, t_, s8_)                                                                                    \
  __builtin_bit_cast(bf16x8, (img_)[((t_) * 16 + l15) * 32 + ((((s8_) * 4 + kq) & 16) | ((((s8_) * 4 + kq) & 15) ^ l15))])
// LDS-DMA of 1 KiB pieces of a packed image: piece i_ = 64 uint4 starting at src_[512 i_] (wave-uniform) -> LDS byte address
// dst_ + 8192 i_.  Scalar base per image + one per-lane offset register (lane * 16, the piece offset added per instruction from four
// loop-invariant registers): with the whole chunk index in the vector offset the compiler kept a dozen loop-invariant offset registers
// alive and spilled around the 256-VGPR loop; with it in precomputed scalar bases it ran out of SGPRs instead.  All pieces of an image
// go out in ONE statement: M0 is saved and restored once, not per piece.
#define MLP_GLDS_4(src_, dst_)                                                                                     \
  do {                                                                                                             \
    unsigned keep__;                                                                                               \
    asm volatile("s_mov_b32 %0, m0\n\t"                                                                            \
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"                               \
                 "s_add_u32 m0, %6, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"                       \
                 "s_add_u32 m0, %6, 0x4000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"                       \
                 "s_add_u32 m0, %6, 0x6000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"                       \
                 "s_mov_b32 m0, %0"                                                                                \
                 : "=&s"(keep__)                                                                                   \
                 : "v"(lane16), "v"(lane16 + 8192u), "v"(lane16 + 16384u), "v"(lane16 + 24576u), "s"(src_), "s"(dst_) \
                 : "memory", "scc");                                                                               \
  } while (0)
#define MLP_GLDS_2(src_, dst_)                                                                                     \
  do {                                                                                                             \
    unsigned keep__;                                                                                               \
    asm volatile("s_mov_b32 %0, m0\n\t"                                                                            \
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"                               \
                 "s_add_u32 m0, %4, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"                       \
                 "s_mov_b32 m0, %0"                                                                                \
                 : "=&s"(keep__)                                                                                   \
                 : "v"(lane16), "v"(lane16 + 8192u), "s"(src_), "s"(dst_)                                          \
                 : "memory", "scc");                                                                               \
  } while (0)
// in-kernel stamps exist in the diagnostic build only (-DMLP_STAMPS, tools/attn_stamps.sh builds it into libtitok_hip_stamps.so for
// tools/mlp_ablate.py): in the product build they cost two s_memtime, a compare / saveexec / branch each per panel iteration
#ifdef MLP_STAMPS
#define MLP_STAMP()                                                                                                \
  do {                                                                                                             \
    if (p.stamps && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && n_stamp < 64)                               \
      p.stamps[(wave >> 2) * 64 + n_stamp++] = (long long)__builtin_readcyclecounter();                            \
  } while (0)
#else
#define MLP_STAMP() do { (void)n_stamp; } while (0)
#endif

// One wave's share of one block of TPB 16-token tiles (block-global tile indices): P1 of tiles [P1F, P1F+T1), P2 of tiles
// [P2F, P2F+T2).  P1F / P2F are wave-uniform run-time values: waves with the same (T1, T2) share one instantiation.
template <int TPB, int T1, int T2, bool GELU_FIRST, bool KEEL, bool FRONT, bool BACK>
__device__ __forceinline__ void mlp_wave(const MlpDev& p, uint4* l12, uint4* l3, uint4* hb, const float* gl, int tile, int wave, int lane,
                                         int P1F, int P2F, int& n_stamp) {
  constexpr int T1A = T1 > 0 ? T1 : 1, T2A = T2 > 0 ? T2 : 1;   // array extents (a role may have no P1 or no P2 tile)
  const int l15 = lane & 15, kq = lane >> 4;
  const int np = p.I / 32;
  // LDS-DMA of one image by all 8 waves: wave w copies KiB blocks w, w+8, ... (64 lanes x 16 B, lane-linear on both
  // sides); asm with a SCALAR base + one per-lane byte offset: no 64-bit per-lane pointers live across the panel loop
  const uint32_t lane16 = lane * 16;
  const uint32_t lds_l12 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)l12;
  const uint32_t lds_l3 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)l3;
#define GLDS_W12(pn_, buf_)                                                                                        \
  do {                                                                                                             \
    const uint4* src__ = p.pack + ((size_t)(pn_) * MLP_PANEL_CHUNKS + wave * 64);                                  \
    const uint32_t dst__ = lds_l12 + ((buf_) * MLP_W12_CHUNKS + wave * 64) * 16;                                   \
    MLP_GLDS_4(src__, dst__);                                                                                      \
  } while (0)
#define GLDS_W3(pn_, buf_)                                                                                         \
  do {                                                                                                             \
    const uint4* src__ = p.pack + ((size_t)(pn_) * MLP_PANEL_CHUNKS + MLP_W12_CHUNKS + wave * 64);                 \
    const uint32_t dst__ = lds_l3 + ((buf_) * MLP_W3_CHUNKS + wave * 64) * 16;                                     \
    MLP_GLDS_2(src__, dst__);                                                                                      \
  } while (0)

  const int tok0 = tile * (16 * TPB) + l15;   // token of block tile g: tok0 + 16 g
  // explicitly global (address space 1) byte pointers: accesses are then global_* with an SGPR base, never flat_*
  typedef __attribute__((address_space(1))) char gchar;
  typedef __attribute__((address_space(1))) const char gcchar;
  gcchar* const gx = (gcchar*)p.x;
  gcchar* const gao = (gcchar*)p.ao;
  gchar* const gy = (gchar*)p.y;

  MLP_STAMP();
  __syncthreads();   // every wave is done with the previous tile's LDS contents

  bf16x8 bfr[T1A][8];   // x rows of the P1 tiles as B fragments, k order per 32-step: {4kq..+3, 16+4kq..+3}
  float rstd[T1A];
  if (FRONT) {
    // ---- fused front: x' = [RMSNorm](fa*x + ao Wo^T)[*fg] for the P1 tiles (the wave then owns all 256 features of its
    // tokens in C layout, which IS the B-fragment layout above); Wo streams through the W12 buffers as 4 panel images ----
#define GLDS_WO(wp_, buf_)                                                                                         \
  do {                                                                                                             \
    const uint4* src__ = p.pack + ((size_t)np * MLP_PANEL_CHUNKS + (wp_) * MLP_W12_CHUNKS + wave * 64);            \
    const uint32_t dst__ = lds_l12 + ((buf_) * MLP_W12_CHUNKS + wave * 64) * 16;                                   \
    MLP_GLDS_4(src__, dst__);                                                                                      \
  } while (0)
    GLDS_WO(0, 0);
    bf16x8 abf[T1A][8];
#pragma unroll
    for (int j = 0; j < T1; ++j) {
      const int t = tok0 + 16 * (P1F + j);
      const int tc = t < p.M ? t : p.M - 1;
      const uint32_t aoff = ((uint32_t)tc * (uint32_t)p.ldao + kq * 8) * 2u;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) abf[j][s8] = *(const __attribute__((address_space(1))) bf16x8*)(gao + aoff + s8 * 64);
    }
    f32x4 facc[16][T1A];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
#pragma unroll
    for (int wp = 0; wp < 4; ++wp) {
      if (wp + 1 < 4) GLDS_WO(wp + 1, (wp + 1) & 1);
      const uint4* img = l12 + (wp & 1) * MLP_W12_CHUNKS;
      bf16x8 a[2][4];
      if constexpr (T1 > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[0][i] = MLP_LD12(img, i, 0);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        if (s8 + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) a[(s8 + 1) & 1][i] = MLP_LD12(img, i, s8 + 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < T1; ++j)
            facc[4 * wp + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s8 & 1][i], abf[j][s8],
                                                                           s8 ? facc[4 * wp + i][j] : (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (s8 + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * T1A, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
#undef GLDS_WO
    GLDS_W12(0, 0);
    // residual, row statistics, gain; the bf16-rounded x' becomes the B fragments of phase 1, and after the panel loop the very
    // same registers are handed to the P2 owner through LDS as the residual of its epilogue (x' never goes to global memory)
#pragma unroll
    for (int j = 0; j < T1; ++j) {
      const int t = tok0 + 16 * (P1F + j);
      const int tc = t < p.M ? t : p.M - 1;
      // scalar base + 32-bit per-lane byte offsets everywhere below: 64-bit row pointers in VGPRs get spilled in this kernel
      // and every scratch reload comes with a vmcnt(0) that serialises the stores / drains the LDS-DMA in flight
      const uint32_t roff = ((uint32_t)tc * (uint32_t)p.ldx + kq * 4) * 2u;
      float ss = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        facc[m][j] += p.front_alpha * cvt4(*(const __attribute__((address_space(1))) bf16x4*)(gx + roff + m * 32));
#pragma unroll
        for (int e = 0; e < 4; ++e) ss = fmaf(facc[m][j][e], facc[m][j][e], ss);
      }
      float scale = 1.0f;
      if (p.front_gain) {
        ss = quad16_sum(ss);
        scale = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
      }
      float ss2 = 0.f;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        bf16x4 q[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int m = 2 * s8 + u;
          f32x4 v = facc[m][j];
          if (p.front_gain) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gl + 256 + m * 16 + kq * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * scale * g[e];
          }
          q[u] = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float r = (float)q[u][e];
            ss2 = fmaf(r, r, ss2);
          }
        }
        bfr[j][s8] = (bf16x8){q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3]};
      }
      ss2 = quad16_sum(ss2);
      rstd[j] = 1.0f / sqrtf(ss2 * (1.0f / 256.0f) + p.eps);
    }
  } else {
    GLDS_W12(0, 0);
    // ---- P1 tiles: B fragments + folded pre-norm rstd ----
#pragma unroll
    for (int j = 0; j < T1; ++j) {
      const int t = tok0 + 16 * (P1F + j);
      const int tc = t < p.M ? t : p.M - 1;
      const uint32_t xoff = ((uint32_t)tc * (uint32_t)p.ldx + kq * 4) * 2u;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const u32x2_t lo = *(const __attribute__((address_space(1))) u32x2_t*)(gx + xoff + s8 * 64);
        const u32x2_t hi = *(const __attribute__((address_space(1))) u32x2_t*)(gx + xoff + (s8 * 64 + 32));
        bfr[j][s8] = __builtin_bit_cast(bf16x8, (u32x4_t){lo.x, lo.y, hi.x, hi.y});
      }
    }
#pragma unroll
    for (int j = 0; j < T1; ++j) {
      float ss = 0.f;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (float)bfr[j][s8][e];
          ss = fmaf(v, v, ss);
        }
      ss = quad16_sum(ss);
      rstd[j] = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
    }
  }
  // ---- P2 tiles: y accumulators ----
  f32x4 out[16][T2A];
#pragma unroll
  for (int m = 0; m < 16; ++m)
#pragma unroll
    for (int j = 0; j < T2; ++j) out[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  MLP_STAMP();

  for (int it = 0; it <= np; ++it) {
    if (it + 1 < np && !(p.debug & 2)) GLDS_W12(it + 1, (it + 1) & 1);
    if (it < np && (!(p.debug & 2) || it == 0)) GLDS_W3(it, it & 1);

    f32x4 acc1[4][T1A];   // [x 0..15, x 16..31, gate 0..15, gate 16..31][P1 token tile]
    if (T1 > 0 && it < np) {
      // ---- P1 MFMAs of panel `it`: A fragments double-buffered in registers, reads pinned ahead of their MFMAs ----
      const uint4* img = l12 + (it & 1) * MLP_W12_CHUNKS;
      bf16x8 a[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[0][i] = MLP_LD12(img, i, 0);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        if (s8 + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) a[(s8 + 1) & 1][i] = MLP_LD12(img, i, s8 + 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < T1; ++j)
            acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s8 & 1][i], bfr[j][s8],
                                                                  s8 ? acc1[i][j] : (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (s8 + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * T1A, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // GELU_FIRST swaps the order of the two remaining parts (measured: both waves P2-before-GEGLU is the faster setting -
    // the P2 MFMAs cover the latency of the P1 results the GEGLU reads)
    if (!GELU_FIRST) {
      if (T2 > 0 && it >= 1 && !(p.debug & 16)) {
        // ---- P2 MFMAs of panel it-1: y^T += W3slice h^T ----
        const int pn = it - 1;
        const uint4* w3l = l3 + (pn & 1) * MLP_W3_CHUNKS + kq * 256 + l15;
        const uint4* hsrc = hb + ((pn & 1) * TPB + P2F) * 64 + lane;
        bf16x8 hf[T2A];
  #pragma unroll
        for (int j = 0; j < T2; ++j) hf[j] = __builtin_bit_cast(bf16x8, hsrc[j * 64]);
        bf16x8 w3f[4];
  #pragma unroll
        for (int r = 0; r < 4; ++r) w3f[r] = __builtin_bit_cast(bf16x8, w3l[r * 16]);
  #pragma unroll
        for (int m = 0; m < 16; ++m) {
  #pragma unroll
          for (int j = 0; j < T2; ++j) out[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[m & 3], hf[j], out[m][j], 0, 0, 0);
          if (m + 4 < 16) w3f[m & 3] = __builtin_bit_cast(bf16x8, w3l[(m + 4) * 16]);
          __builtin_amdgcn_sched_group_barrier(0x008, T2A, 0);
          if (m + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (T1 > 0 && it < np) {
      // ---- GEGLU of panel `it` in registers -> one B fragment of the second product per token tile ----
      uint4* hdst = hb + ((it & 1) * TPB + P1F) * 64 + lane;
#pragma unroll
      for (int j = 0; j < T1; ++j) {
        float gg[8], xx[8], hh[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            gg[4 * i + e] = acc1[2 + i][j][e] * rstd[j];
            xx[4 * i + e] = acc1[i][j][e] * rstd[j];
          }
        if (p.debug & 4) {      // diagnostic (tools/mlp_ablate.py): no GEGLU arithmetic, garbage results
#pragma unroll
          for (int k = 0; k < 8; ++k) hh[k] = gg[k] + xx[k];
        } else {
          geglu_fast8(gg, xx, hh);
        }
        uint32_t hw[4];
        typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bf16x2_t q = {(bf16_t)hh[2 * k], (bf16_t)hh[2 * k + 1]};
          hw[k] = __builtin_bit_cast(uint32_t, q);
        }
        hdst[j * 64] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      }
    }
    if (GELU_FIRST) {
      if (T2 > 0 && it >= 1 && !(p.debug & 16)) {
        // ---- P2 MFMAs of panel it-1: y^T += W3slice h^T ----
        const int pn = it - 1;
        const uint4* w3l = l3 + (pn & 1) * MLP_W3_CHUNKS + kq * 256 + l15;
        const uint4* hsrc = hb + ((pn & 1) * TPB + P2F) * 64 + lane;
        bf16x8 hf[T2A];
  #pragma unroll
        for (int j = 0; j < T2; ++j) hf[j] = __builtin_bit_cast(bf16x8, hsrc[j * 64]);
        bf16x8 w3f[4];
  #pragma unroll
        for (int r = 0; r < 4; ++r) w3f[r] = __builtin_bit_cast(bf16x8, w3l[r * 16]);
  #pragma unroll
        for (int m = 0; m < 16; ++m) {
  #pragma unroll
          for (int j = 0; j < T2; ++j) out[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[m & 3], hf[j], out[m][j], 0, 0, 0);
          if (m + 4 < 16) w3f[m & 3] = __builtin_bit_cast(bf16x8, w3l[(m + 4) * 16]);
          __builtin_amdgcn_sched_group_barrier(0x008, T2A, 0);
          if (m + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (it >= 4 && it < 8) MLP_STAMP();     // work done
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the next images is in LDS
    __syncthreads();
    if (it >= 4 && it < 8) MLP_STAMP();     // barrier passed
  }
  MLP_STAMP();

  // ---- FRONT: hand the x' rows of the P1 tiles to their P2 owners.  The weight images are dead after the loop's last barrier and
  // the 96 KiB they occupied hold exactly the 4 x NT token tiles of the block at 8 KiB each; a lane of the P2 wave needs, for the
  // m-tiles 2 s8 and 2 s8 + 1, the very 8 values the same lane of the P1 wave holds in bfr[.][s8] (the C layout of the second
  // product IS the B-fragment layout of the first): one 16-byte vector per (tile, s8), lane to same lane, no conflicts. ----
  uint4* const xres = l12;      // [TPB tiles][8][64 lanes] (<= 96 KiB: 12 tiles); l3 follows l12 in the block's LDS
  if (FRONT) {
#pragma unroll
    for (int j = 0; j < T1; ++j)
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) xres[((P1F + j) * 8 + s8) * 64 + lane] = __builtin_bit_cast(uint4, bfr[j][s8]);
    __syncthreads();
  }

  // ---- epilogue of the P2 tiles: y = alpha*x + acc ; x_new = KEEL ? RMSNorm(y)*gain : y ----
  bf16x8 xq[T2A][8];
  float rstdq[T2A];
#pragma unroll
  for (int j = 0; j < T2; ++j) {
    const int t = tok0 + 16 * (P2F + j);
    const bool tv = t < p.M;
    const int tc = tv ? t : p.M - 1;
    // FRONT: the residual is the x' its partner wave left in LDS; otherwise the input rows (scalar base + 32-bit byte offsets)
    const uint32_t roff = ((uint32_t)tc * (uint32_t)p.ldx + kq * 4) * 2u;
    float ss = 0.f;
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      f32x4 rv0, rv1;
      if (FRONT) {
        const bf16x8 rb = __builtin_bit_cast(bf16x8, xres[((P2F + j) * 8 + s8) * 64 + lane]);
        rv0 = (f32x4){(float)rb[0], (float)rb[1], (float)rb[2], (float)rb[3]};
        rv1 = (f32x4){(float)rb[4], (float)rb[5], (float)rb[6], (float)rb[7]};
      } else {
        rv0 = cvt4(*(const __attribute__((address_space(1))) bf16x4*)(gx + roff + (2 * s8) * 32));
        rv1 = cvt4(*(const __attribute__((address_space(1))) bf16x4*)(gx + roff + (2 * s8 + 1) * 32));
      }
      out[2 * s8][j] += p.alpha * rv0;
      out[2 * s8 + 1][j] += p.alpha * rv1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ss = fmaf(out[2 * s8][j][e], out[2 * s8][j][e], ss);
        ss = fmaf(out[2 * s8 + 1][j][e], out[2 * s8 + 1][j][e], ss);
      }
    }
    float scale = 1.0f;
    if (KEEL) {
      ss = quad16_sum(ss);
      scale = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
    }
    const bool odd = kq & 1;
    const uint32_t yoff = ((uint32_t)tc * (uint32_t)p.ldy + (odd ? 16 + kq * 4 - 4 : kq * 4)) * 2u;   // + 64 bytes per ip
    float ssq = 0.f;
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
      f32x4 y0 = out[i0][j], y1 = out[i1][j];
      if (KEEL) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gl + i0 * 16 + kq * 4);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gl + i1 * 16 + kq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = y0[e] * scale * g0[e]; y1[e] = y1[e] * scale * g1[e]; }
      }
      const bf16x4 q0 = {(bf16_t)y0[0], (bf16_t)y0[1], (bf16_t)y0[2], (bf16_t)y0[3]};
      const bf16x4 q1 = {(bf16_t)y1[0], (bf16_t)y1[1], (bf16_t)y1[2], (bf16_t)y1[3]};
      const uint2 p0 = __builtin_bit_cast(uint2, q0), p1 = __builtin_bit_cast(uint2, q1);
      if (BACK) {   // the rounded new rows in C layout are the B fragments of the next projection
        xq[j][ip] = __builtin_bit_cast(bf16x8, make_uint4(p0.x, p0.y, p1.x, p1.y));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r0 = (float)q0[e], r1 = (float)q1[e];
          ssq = fmaf(r0, r0, ssq);
          ssq = fmaf(r1, r1, ssq);
        }
      }
      const uint4 ov = xchg16_pair(p0, p1);     // even kq: (own p0, partner's p0), odd kq: (partner's p1, own p1) - one VALU op per dword
      if (tv && !(p.debug & 1)) *(__attribute__((address_space(1))) u32x4_t*)(gy + yoff + ip * 64) = (u32x4_t){ov.x, ov.y, ov.z, ov.w};
    }
    if (BACK) {
      ssq = quad16_sum(ssq);
      rstdq[j] = 1.0f / sqrtf(ssq * (1.0f / 256.0f) + p.eps);
    }
  }
  MLP_STAMP();

  if (BACK) {
    // ---- fused back: the next layer's qkv = rope(RMSNorm(x_new) Wqkv'^T) for the P2 tiles, straight from the registers
    // the epilogue just produced; Wqkv' streams through the W12 buffers as nqp more panel images (all 8 waves copy).
    // Explicitly-global row pointers + immediates, as everywhere in this kernel. ----
#define GLDS_WQ(qp_, buf_)                                                                                         \
  do {                                                                                                             \
    const uint4* src__ = p.pack + ((size_t)np * MLP_PANEL_CHUNKS + MLP_WO_CHUNKS + (qp_) * MLP_W12_CHUNKS + wave * 64); \
    const uint32_t dst__ = lds_l12 + ((buf_) * MLP_W12_CHUNKS + wave * 64) * 16;                                   \
    MLP_GLDS_4(src__, dst__);                                                                                      \
  } while (0)
    __syncthreads();      // every wave has read its residuals out of the image buffers
    GLDS_WQ(0, 0);
    gchar* const gq = (gchar*)p.qkv;
    gcchar* const grope = (gcchar*)p.rope_cs;
    // rotary (cos, sin) of this lane's features: every rotary panel is one 64-wide head, so they depend on the token only
    f32x2 rc[4][T2A], rs[4][T2A];
    uint32_t qoff[T2A];
    bool qv[T2A];
#pragma unroll
    for (int j = 0; j < T2; ++j) {
      const int t = tok0 + 16 * (P2F + j);
      qv[j] = t < p.M;
      const int tc = qv[j] ? t : p.M - 1;
      qoff[j] = ((uint32_t)tc * (uint32_t)p.ldq + kq * 4) * 2u;
      gcchar* cs = grope + ((uint32_t)tc * 64u + kq * 2) * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rc[i][j] = *(const __attribute__((address_space(1))) f32x2*)(cs + i * 32);
        rs[i][j] = *(const __attribute__((address_space(1))) f32x2*)(cs + 128 + i * 32);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int qp = 0; qp < p.nqp; ++qp) {
      if (qp + 1 < p.nqp) GLDS_WQ(qp + 1, (qp + 1) & 1);
      const uint4* img = l12 + (qp & 1) * MLP_W12_CHUNKS;
      f32x4 qa[4][T2A];
      bf16x8 a[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[0][i] = MLP_LD12(img, i, 0);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        if (s8 + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) a[(s8 + 1) & 1][i] = MLP_LD12(img, i, s8 + 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < T2; ++j)
            qa[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s8 & 1][i], xq[j][s8], s8 ? qa[i][j] : (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (s8 + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * T2A, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // wait + barrier BEFORE this panel's stores are issued: the vmcnt(0) then covers the next image's DMA and the
      // previous panel's stores (a whole MFMA phase old), never the stores just issued
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      const int f_first = qp * 64;
      const bool rot = f_first < p.rope_q_end || (f_first >= p.rope_k_begin && f_first < p.rope_k_end);
#pragma unroll
      for (int j = 0; j < T2; ++j) {
        // 16-byte stores: lanes kq / kq^1 exchange one 4-feature group (as in the main epilogue), so a row gets 64-byte runs
        const bool odd = kq & 1;
        gchar* const qrow = gq + (qoff[j] - (uint32_t)(kq * 4) * 2u) + (uint32_t)(f_first + (odd ? 16 + kq * 4 - 4 : kq * 4)) * 2u;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          uint2 pk[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = 2 * ip + u;
            f32x4 v = qa[i][j] * rstdq[j];
            if (rot)
              v = (f32x4){v[0] * rc[i][j].x - v[1] * rs[i][j].x, v[0] * rs[i][j].x + v[1] * rc[i][j].x,
                          v[2] * rc[i][j].y - v[3] * rs[i][j].y, v[2] * rs[i][j].y + v[3] * rc[i][j].y};
            const bf16x4 q = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            pk[u] = __builtin_bit_cast(uint2, q);
          }
          const uint4 ox = xchg16_pair(pk[0], pk[1]);
          const u32x4_t ov = {ox.x, ox.y, ox.z, ox.w};
          if (qv[j]) *(__attribute__((address_space(1))) u32x4_t*)(qrow + ip * 64) = ov;
        }
      }
    }
#undef GLDS_WQ
    MLP_STAMP();
  }
#undef GLDS_W12
#undef GLDS_W3
}

// TILES = 16-token tiles per block.  4 NT (NT = 1, 2, 3): the pair layout described above.  9 (144 tokens): the benchmark batch
// (36 864 tokens) is then 256 blocks - one per CU instead of 192 - and the nine tiles are dealt over the eight waves so that the
// longest wave is shorter than in the pair layout (lone-wave cost of a P1 tile ~ 3.5 P2 tiles: 32 MFMAs + the GEGLU against 16 MFMAs):
//     wave 0..3: P1 of tiles {2w, 2w+1}      wave 4: P2 of {0,1,2}   wave 5: P2 of {3,4,5}   wave 6: P1 of {8} + P2 of {6}   wave 7: P2 of {7,8}
// (SIMD s runs waves s and s+4).  Four role programs instead of two.
template <int TILES, bool KEEL, bool FRONT, bool BACK>
__global__ __launch_bounds__(512, 1) void k_mlp256(MlpDev p) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* const l12 = smem;                                 // [2][MLP_W12_CHUNKS]
  uint4* const l3 = smem + 2 * MLP_W12_CHUNKS;             // [2][MLP_W3_CHUNKS]
  uint4* const hb = l3 + 2 * MLP_W3_CHUNKS;                // [2][TILES][64 lanes]
  float* const gl = reinterpret_cast<float*>(hb + 2 * TILES * 64);   // [256] post_gain, [256] front_gain (read per m-tile by
                                                                      // low-latency LDS loads instead of serialised global loads)
  if (threadIdx.x < 256) {
    gl[threadIdx.x] = p.post_gain ? p.post_gain[threadIdx.x] : 1.0f;
    gl[256 + threadIdx.x] = (FRONT && p.front_gain) ? p.front_gain[threadIdx.x] : 1.0f;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: DMA bases / role branch stay in SGPRs
  int n_stamp = 0;
  // one tile per block (grid = n_tiles; with more tiles than CUs the dispatcher queues the rest - one block fits a CU): inside a
  // persistent tile loop the compiler hoisted lane constants of every phase to the top and spilled them around the 256-VGPR body
  const int tile = blockIdx.x;
  if constexpr (TILES == 9) {
    if (wave < 4) mlp_wave<9, 2, 0, false, KEEL, FRONT, BACK>(p, l12, l3, hb, gl, tile, wave, lane, 2 * wave, 0, n_stamp);
    else if (wave < 6) mlp_wave<9, 0, 3, false, KEEL, FRONT, BACK>(p, l12, l3, hb, gl, tile, wave, lane, 0, 3 * (wave - 4), n_stamp);
    else if (wave == 6) mlp_wave<9, 1, 1, false, KEEL, FRONT, BACK>(p, l12, l3, hb, gl, tile, wave, lane, 8, 6, n_stamp);
    else mlp_wave<9, 0, 2, false, KEEL, FRONT, BACK>(p, l12, l3, hb, gl, tile, wave, lane, 0, 7, n_stamp);
  } else {
    constexpr int NT = TILES / 4;
    constexpr int TA = NT - NT / 2;   // P1 tiles of the first wave of a pair (2 of 3, 1 of 2)
    const int g0 = (wave & 3) * NT;   // first tile of the wave pair
    if (wave < 4) mlp_wave<TILES, TA, NT - TA, false, KEEL, FRONT, BACK>(p, l12, l3, hb, gl, tile, wave, lane, g0, g0 + TA, n_stamp);
    else mlp_wave<TILES, NT - TA, TA, false, KEEL, FRONT, BACK>(p, l12, l3, hb, gl, tile, wave, lane, g0 + TA, g0, n_stamp);
  }
}
#undef MLP_LD12
#undef MLP_GLDS_4
#undef MLP_GLDS_2
#undef MLP_STAMP

bool ttvk_mlp_fused_supported(int dtype, int width, int inner) { return dtype == TTV_BF16 && width == 256 && inner % 32 == 0 && inner > 0; }

int64_t ttvk_mlp_pack_bytes(int inner, int next_qkv_rows) {
  if (inner <= 0 || inner % 32 != 0 || next_qkv_rows < 0) return 0;
  return ((int64_t)(inner / 32) * MLP_PANEL_CHUNKS + MLP_WO_CHUNKS + (int64_t)((next_qkv_rows + 63) / 64) * MLP_W12_CHUNKS) * 16;
}

int ttvk_mlp_pack(const void* w12_folded, const void* w3, const void* wo, const void* next_qkv_folded, int next_qkv_rows, int inner,
                  void* packed, hipStream_t s) {
  TTV_CHECK_ARG(w12_folded && w3 && packed, "mlp_pack: null buffer");
  TTV_CHECK_ARG(inner > 0 && inner % 32 == 0, "mlp_pack: inner %% 32");
  TTV_CHECK_ARG((next_qkv_folded != nullptr) == (next_qkv_rows > 0), "mlp_pack: next_qkv weight and row count must come together");
  TTV_CHECK_ARG(((uintptr_t)w12_folded % 16 == 0) && ((uintptr_t)packed % 16 == 0) && ((uintptr_t)wo % 16 == 0) &&
                    ((uintptr_t)next_qkv_folded % 16 == 0), "mlp_pack: 16-byte alignment");
  const int total = inner / 32 * MLP_PANEL_CHUNKS + MLP_WO_CHUNKS + (next_qkv_rows + 63) / 64 * MLP_W12_CHUNKS;
  hipLaunchKernelGGL(k_mlp_pack, dim3(ttv_cdiv(total, 256)), dim3(256), 0, s, (const bf16_t*)w12_folded, (const bf16_t*)w3,
                     (const bf16_t*)wo, (const bf16_t*)next_qkv_folded, next_qkv_rows, inner, (uint4*)packed);
  TTV_CHECK_LAUNCH("mlp_pack");
  return TTV_OK;
}

int ttvk_mlp_fused(const void* ao, int ldao, const float* front_gain, float front_alpha, const void* x, int ldx, const void* packed,
                   int inner, void* y, int ldy, const float* post_gain, float alpha, float eps, int M, const MlpNextQkv* nq, hipStream_t s) {
  if (M == 0) return TTV_OK;
  TTV_CHECK_ARG(x && packed && y, "mlp_fused: null buffer");
  TTV_CHECK_ARG(inner % 32 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (!ao || ldao % 8 == 0), "mlp_fused: inner %% 32, leading dims %% 8");
  TTV_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)packed % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)ao % 16 == 0),
                "mlp_fused: 16-byte alignment");
  MlpDev d;
  d.x = (const bf16_t*)x; d.ldx = ldx; d.pack = (const uint4*)packed; d.I = inner;
  d.ao = (const bf16_t*)ao; d.ldao = ldao; d.front_gain = front_gain; d.front_alpha = front_alpha;
  d.y = (bf16_t*)y; d.ldy = ldy; d.post_gain = post_gain; d.alpha = alpha; d.eps = eps; d.M = M; d.debug = g_ttv_debug; d.stamps = g_ttv_stamps;
  d.qkv = nullptr; d.ldq = 0; d.rope_cs = nullptr; d.nqp = 0; d.rope_q_end = d.rope_k_begin = d.rope_k_end = 0;
  if (nq) {
    TTV_CHECK_ARG(nq->qkv && nq->rope_cs && nq->rows > 0 && nq->rows % 64 == 0 && nq->ld >= nq->rows && nq->ld % 4 == 0 &&
                      (uintptr_t)nq->qkv % 8 == 0, "mlp_fused: bad next-qkv arguments (rows %% 64, ld %% 4, 8-byte aligned)");
    TTV_CHECK_ARG(nq->rope_q_end % 64 == 0 && nq->rope_k_begin % 64 == 0 && nq->rope_k_end % 64 == 0, "mlp_fused: rotary ranges must be whole heads");
    d.qkv = (bf16_t*)nq->qkv; d.ldq = nq->ld; d.rope_cs = nq->rope_cs; d.nqp = nq->rows / 64;
    d.rope_q_end = nq->rope_q_end; d.rope_k_begin = nq->rope_k_begin; d.rope_k_end = nq->rope_k_end;
  }
  // 16-token tiles per block: 4, 8, 12 (wave pairs with 1, 2, 3 tiles each) or 9 (the eight-wave deal of k_mlp256).  The choice
  // minimises (rounds over the 256 CUs) x (cost of one block): a block costs about 16 + 19 NT us in the pair layout (measured: 54 us at
  // NT = 2, 62 us at NT = 3) - small batches (the reference trains under a 6144-token budget, configs/tiny.yaml:65) take 64-token
  // blocks on many CUs rather than a few CUs with fat blocks - and 9-tile blocks cost less than 12-tile ones and fill all 256 CUs at
  // the benchmark batch (36 864 tokens = 256 x 144).  The fused next-QKV phase exists for the pair layouts only.
  const int cus = 256;
  int tiles = 4;
  long best = -1;
  for (int c = 1; c <= 3; ++c) {
    const long cost = (long)ttv_cdiv(ttv_cdiv(M, 64 * c), cus) * (16 + 19 * c);
    if (best < 0 || cost < best) { best = cost; tiles = 4 * c; }
  }
  if (!nq && !(g_ttv_debug & 32)) {      // debug bit5: pair layouts only (A/B)
    const long cost9 = (long)ttv_cdiv(ttv_cdiv(M, 144), cus) * (16 + 14 * 3);
    if (cost9 < best) { best = cost9; tiles = 9; }
  }
  if ((g_ttv_debug & 512) && !nq) tiles = 9;   // debug bit9 forces the 9-tile deal (tests)
  if (g_ttv_debug & 8) tiles = 8;      // debug bit3 forces the 2-tile pair variant
  if (g_ttv_debug & 64) tiles = 4;     // debug bit6 forces the 1-tile pair variant
  d.n_tiles = ttv_cdiv(M, 16 * tiles);
  const int grid = d.n_tiles;
  const size_t smem = (size_t)(2 * MLP_W12_CHUNKS + 2 * MLP_W3_CHUNKS + 2 * tiles * 64) * sizeof(uint4) + 2048;   // 96 KiB + 2 KiB per token tile + gains
  TtvProfScope prof(TTV_KC_GEMM_GEGLU, s);
#define LAUNCH_MLP(TILES_, KEEL_, FRONT_, BACK_)                                                                    \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)k_mlp256<TILES_, KEEL_, FRONT_, BACK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((k_mlp256<TILES_, KEEL_, FRONT_, BACK_>), dim3(grid), dim3(512), smem, s, d);                \
  } while (0)
  // the fused back exists only together with the fused front (the layer-tail use): keeps the instantiation count down
#define LAUNCH_MLP_F(TILES_, KEEL_)                                                                                 \
  do {                                                                                                              \
    if (ao && nq) LAUNCH_MLP(TILES_, KEEL_, true, true);                                                            \
    else if (ao) LAUNCH_MLP(TILES_, KEEL_, true, false);                                                            \
    else LAUNCH_MLP(TILES_, KEEL_, false, false);                                                                   \
  } while (0)
#define LAUNCH_MLP_9(KEEL_)                                                                                         \
  do {                                                                                                              \
    if (ao) LAUNCH_MLP(9, KEEL_, true, false);                                                                      \
    else LAUNCH_MLP(9, KEEL_, false, false);                                                                        \
  } while (0)
  TTV_CHECK_ARG(!nq || ao, "mlp_fused: the next-qkv part needs the out_proj front (ttv_layer_tail_fused)");
  if (post_gain) {
    if (tiles == 9) LAUNCH_MLP_9(true); else if (tiles == 12) LAUNCH_MLP_F(12, true); else if (tiles == 8) LAUNCH_MLP_F(8, true); else LAUNCH_MLP_F(4, true);
  } else {
    if (tiles == 9) LAUNCH_MLP_9(false); else if (tiles == 12) LAUNCH_MLP_F(12, false); else if (tiles == 8) LAUNCH_MLP_F(8, false); else LAUNCH_MLP_F(4, false);
  }
#undef LAUNCH_MLP_9
#undef LAUNCH_MLP_F
#undef LAUNCH_MLP
  TTV_CHECK_LAUNCH("mlp_fused");
  return TTV_OK;
}

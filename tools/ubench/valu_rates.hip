// Issue cost of v_exp_f32 / v_add_f32 / v_cvt_pk_bf16_f32 / v_mfma_f32_32x32x16_bf16 for 1..4 waves per SIMD, alone and mixed.
// Diagnostic; hipcc --offload-arch=gfx950.
//
// Metric (round 3; the round-2 version printed mean(per-wave t1 - t0) / waves, which under oldest-first arbitration is 2/3 of the
// truth at three waves and read "20 cycles per MFMA per SIMD", above the part's peak): one block per CU of 4 w waves, i.e. w waves on
// each SIMD; every wave stamps s_memtime after a block barrier (t0) and after its loop (t1); the block's MAKESPAN is
// max(t1) - min(t0) over its waves, in which each SIMD ran w x iters iterations.  KIND >= 20 rows no longer run the KIND-10 loop
// first (round 2: "s_barrier 921.9" was 905 cycles of that loop + 17).  Printed: the median over the 256 blocks of
// makespan / (iters x w x per_iter) = cycles per instruction (or per iteration) PER SIMD, next to mean(t1 - t0) / (iters x per_iter),
// what one wave waits for its own instruction.  Check: "mfma alone" must read ~32 per SIMD at every occupancy.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  f32x16 acc = {}, acc2 = {};
  float w[16];
  for (int i = 0; i < 16; ++i) w[i] = threadIdx.x * 2e-3f + i;
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f); b[i] = (__bf16)(i * 0.1f); }
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  if (KIND < 10)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
      if (KIND == 1) v[i] = v[i] + 1.25f;
      if (KIND == 2) { v[i] = __builtin_amdgcn_exp2f(v[i]); v[i] = v[i] + 1.25f; }
      if (KIND == 3) { if (i % 4 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); v[i] = __builtin_amdgcn_exp2f(v[i]); }
      if (KIND == 4) { if (i % 4 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }
      if (KIND == 5) v[i] = __builtin_amdgcn_rcpf(v[i]);
      // the attention mix: per MFMA 6 plain + 2 exp2 (KIND 6: two accumulators alternate; 7: one dependent chain; 8: no exp2, 8 plain)
      if (KIND == 6 || KIND == 7) {
        if (i % 2 == 0) {
          if (KIND == 6 && (i & 2)) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        v[i] = __builtin_amdgcn_exp2f(v[i]);                  // 1 exp2 + 3 plain per i, two i per MFMA
        w[i] = w[i] + 1.25f;
        w[(i + 5) & 15] = w[(i + 5) & 15] * 0.75f;
        w[(i + 9) & 15] = w[(i + 9) & 15] + v[(i + 8) & 15];
      }
      if (KIND == 8) {
        if (i % 2 == 0) {
          if (i & 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        w[i] = w[i] + 1.25f;
        w[(i + 5) & 15] = w[(i + 5) & 15] * 0.75f;
        w[(i + 9) & 15] = w[(i + 9) & 15] + 0.5f;
        w[(i + 13) & 15] = w[(i + 13) & 15] * 1.5f;
      }
      if (KIND == 9) {      // 2 accumulators, MFMA only
        if (i % 2 == 0) {
          if (i & 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
      }
    }
  }
  if (KIND >= 10 && KIND < 20) {
    // phase structure of an attention tile, no in-wave interleave: 16 MFMAs (two accumulators), then 96 plain + 32 exp2
    // (KIND 10), or 96 plain only (11), or the MFMAs only (12): do co-resident waves overlap each other's phases?
    for (int it = 0; it < iters; ++it) {
      if (KIND != 13) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (KIND != 12) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (KIND != 11) v[i] = __builtin_amdgcn_exp2f(v[i]);
            w[i] = w[i] + 1.25f;
            w[(i + 5) & 15] = w[(i + 5) & 15] * 0.75f;
            w[(i + 9) & 15] = w[(i + 9) & 15] + 0.5f;
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (KIND >= 50 && KIND < 60) {
    // round 4: ping-pong.  The waves of a block that share a SIMD (wave w and w + 4 of an 8-wave block) run the two phases of the
    // attention tile in ANTI-phase, separated by block barriers: while one issues its 16 MFMAs the other issues its vector work.
    // 50: 96 plain + 32 exp2 (the round-3 mix);  51: 44 plain + 32 exp2 (the loop after round 4: no row maximum, packed sums);
    // 52 / 53: the same two without the barriers and without the anti-phase start (what unsynchronised co-resident waves do)
    // 54 / 55: as 51 / 53 with s_setprio 2 around the VECTOR phase;  56 / 57: with s_setprio 2 around the MFMA phase
    const int grp = (threadIdx.x >> 8) & 1;
    constexpr int NP = (KIND == 50 || KIND == 52) ? 3 : 1;
    constexpr bool SYNC = KIND == 50 || KIND == 51 || KIND == 54 || KIND == 56;
    constexpr int PRIO = (KIND == 54 || KIND == 55) ? 1 : (KIND == 56 || KIND == 57) ? 2 : 0;      // plain per exp2 (+ 12 more below for 51 / 53)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const bool mfma_phase = !SYNC ? (ph == 0) : ((ph ^ grp) == 0);
        if (mfma_phase) {
          if (PRIO == 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          }
          if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        } else {
          if (PRIO == 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              v[i] = __builtin_amdgcn_exp2f(v[i]);
              w[i] = w[i] + 1.25f;
              if (NP == 3) {
                w[(i + 5) & 15] = w[(i + 5) & 15] * 0.75f;
                w[(i + 9) & 15] = w[(i + 9) & 15] + 0.5f;
              }
            }
          if (NP == 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i) w[i] = w[i] * 0.75f;
          }
          if (PRIO == 1) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (SYNC) __builtin_amdgcn_s_barrier();
      }
    }
  }
  if (KIND >= 40 && KIND < 43) {
    // round 4: the same FLOPs on v_mfma_f32_16x16x32_bf16 (two of them per 32x32x16: 32 per iteration), same vector work:
    // 40 = the MFMAs alone, 41 = phases (32 mfma16 | 96 plain + 32 exp2), 42 = interleaved 16 x (2 mfma16, 2 exp2, 6 plain).
    // Does the smaller shape's longer hold on the issue port (8 of 16 cycles instead of 8 of 32) cost a loop with this much vector work?
    typedef __attribute__((ext_vector_type(4))) float f32x4_;
    f32x4_ c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    bf16x8 a2 = a, b2 = b;
    for (int i = 0; i < 8; ++i) { a2[i] = (__bf16)(threadIdx.x * 0.02f + i); b2[i] = (__bf16)(i * 0.3f - 1.f); }
    asm volatile("" : "+v"(a2), "+v"(b2));
    for (int it = 0; it < iters; ++it) {
      if (KIND == 40 || KIND == 41) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, c3, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, c5, 0, 0, 0);
          c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, c7, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (KIND == 41) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              v[i] = __builtin_amdgcn_exp2f(v[i]);
              w[i] = w[i] + 1.25f;
              w[(i + 5) & 15] = w[(i + 5) & 15] * 0.75f;
              w[(i + 9) & 15] = w[(i + 9) & 15] + 0.5f;
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i & 1) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0); }
          else { c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0); }
          v[i] = __builtin_amdgcn_exp2f(v[i]);
          v[(i + 8) & 15] = __builtin_amdgcn_exp2f(v[(i + 8) & 15]);
          w[i] = w[i] + 1.25f; w[(i + 3) & 15] = w[(i + 3) & 15] * 0.75f; w[(i + 5) & 15] = w[(i + 5) & 15] + 0.5f;
          w[(i + 7) & 15] = w[(i + 7) & 15] * 1.5f; w[(i + 9) & 15] = w[(i + 9) & 15] + 0.25f; w[(i + 11) & 15] = w[(i + 11) & 15] * 0.5f;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    for (int e = 0; e < 4; ++e) acc[e] += c0[e] + c1[e] + c2[e] + c3[e] + c4[e] + c5[e] + c6[e] + c7[e];
  }
  if (KIND == 43) {
    // the 32x32x16 counterpart of 42: 16 x (1 mfma32, 2 exp2, 6 plain)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        v[i] = __builtin_amdgcn_exp2f(v[i]);
        v[(i + 8) & 15] = __builtin_amdgcn_exp2f(v[(i + 8) & 15]);
        w[i] = w[i] + 1.25f; w[(i + 3) & 15] = w[(i + 3) & 15] * 0.75f; w[(i + 5) & 15] = w[(i + 5) & 15] + 0.5f;
        w[(i + 7) & 15] = w[(i + 7) & 15] * 1.5f; w[(i + 9) & 15] = w[(i + 9) & 15] + 0.25f; w[(i + 11) & 15] = w[(i + 11) & 15] * 0.5f;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (KIND >= 20 && KIND < 30) {
    // the non-ALU ingredients of an attention tile, per iteration: 20 = 8 ds_read_b128 + 16 ds_read_b64 (consumed by a cheap xor),
    // 21 = 64 dependent s_add, 22 = 4 global_load_lds_dwordx4 (1 KiB each, L2-resident source) with a counted wait,
    // 23 = one s_barrier, 24 = 8 untaken branches on a VALU compare (v_cmp + s_cbranch_vccnz)
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    unsigned sacc = iters;
    unsigned vx = 0;
    for (int it = 0; it < iters; ++it) {
      if (KIND == 20) {
        // all 24 reads are issued before the first result is consumed (round 2 consumed each read at once: 24 exposed LDS
        // latencies per iteration, ~46 cycles each - a latency figure, not an issue cost)
        const char* p0 = smem + (threadIdx.x & 63) * 16;
        uint4 qa[8];
        uint2 qb[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) qa[i] = *reinterpret_cast<const uint4*>(p0 + i * 1024 + (it & 1) * 8192);
#pragma unroll
        for (int i = 0; i < 16; ++i) qb[i] = *reinterpret_cast<const uint2*>(smem + (threadIdx.x & 63) * 8 + i * 512 + 16384 + (it & 1) * 8192);
#pragma unroll
        for (int i = 0; i < 8; ++i) vx ^= qa[i].x ^ qa[i].w;
#pragma unroll
        for (int i = 0; i < 16; ++i) vx ^= qb[i].x ^ qb[i].y;
      }
      if (KIND == 21) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"(it) : "scc");
      }
      if (KIND == 22) {
        const unsigned voff = (threadIdx.x & 63) * 16;
        const float* src = out + (blockIdx.x & 63) * 4096 + (it & 3) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (threadIdx.x >> 6) * 4096 + i * 1024);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff + i * 1024), "s"(src), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      if (KIND == 23) __builtin_amdgcn_s_barrier();
      if (KIND == 24) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (__builtin_amdgcn_ballot_w64(v[i] > 1e30f) != 0ull) { v[i] = __builtin_amdgcn_exp2f(v[i]); vx += 1; }
          v[i] = v[i] * 0.999f;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    v[0] += (float)(vx + sacc);
  }
  if (KIND >= 30 && KIND < 40) {
    // the 64-query-rows-per-wave attention tile as an instruction mix: per iteration 32 MFMAs, each followed by its gap's fillers.
    // 30: 2 exp2 + 4 plain per gap (the full softmax of two 32-row tiles: 64 exp2 + 128 plain);  31: 2 exp2 + 2 plain per gap
    // (row sums on the matrix pipe, no row maximum);  32: as 30 with 4 accumulators round-robin;  33: 1 exp2 + 2 plain per gap
    // (what a head_dim-128 kernel has per MFMA)
    f32x16 acc3 = {}, acc4 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (KIND == 32) {
          if ((i & 3) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
          if ((i & 3) == 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          if ((i & 3) == 2) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
          if ((i & 3) == 3) acc4 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc4, 0, 0, 0);
        } else {
          if (i & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        const int j = i & 15;
        v[j] = __builtin_amdgcn_exp2f(v[j]);
        w[j] = w[j] + 1.25f;
        w[(j + 5) & 15] = w[(j + 5) & 15] * 0.75f;
        if (KIND != 33) v[(j + 8) & 15] = __builtin_amdgcn_exp2f(v[(j + 8) & 15]);
        if (KIND == 30 || KIND == 32) {
          w[(j + 9) & 15] = w[(j + 9) & 15] + 0.5f;
          w[(j + 13) & 15] = w[(j + 13) & 15] * 1.5f;
        }
        // one MFMA and its fillers per scheduling group, in this order
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, KIND == 33 ? 3 : (KIND == 31 ? 4 : 6), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 16; ++i) v[i] += acc3[i] + acc4[i];
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc2[i] + w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) {
    cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t0;
    cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = t1;
  }
}

template <int KIND>
void run(const char* name, int per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 2 * 8192 * 8);
  for (int wps = 1; wps <= 4; ++wps) {      // waves per SIMD: block of 256*wps threads, one block per CU
    const int threads = 256 * wps, iters = 200;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float wall_ms = 0.f;
    hipEventElapsedTime(&wall_ms, e0, e1);
    const int wpb = threads / 64;
    std::vector<long long> h(2 * 256 * wpb);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double own = 0;
    std::vector<double> span(256);
    for (int b = 0; b < 256; ++b) {
      long long lo = h[2 * b * wpb], hi = h[2 * b * wpb + 1];
      for (int w = 0; w < wpb; ++w) {
        const long long a0 = h[2 * (b * wpb + w)], a1 = h[2 * (b * wpb + w) + 1];
        lo = a0 < lo ? a0 : lo; hi = a1 > hi ? a1 : hi; own += (double)(a1 - a0);
      }
      span[b] = (double)(hi - lo);
    }
    own /= 256.0 * wpb;
    std::sort(span.begin(), span.end());
    printf("%-52s %d waves/SIMD: %8.1f cycles per SIMD (block makespan / (iters x waves per SIMD)), %8.1f waited by a wave, launch %7.1f us\n", name, wps,
           span[128] / ((double)iters * per_iter * wps), own / ((double)iters * per_iter), wall_ms * 1e3);
  }
}
int main() {
  if (getenv("PINGPONG")) {         // round 4: anti-phase waves of one block against unsynchronised ones, per iteration (= per unit and wave)
    run<52>("unsynchronised: 16 mfma | 96 plain + 32 exp2 (per iteration)", 1);
    run<50>("ping-pong:      16 mfma | 96 plain + 32 exp2 (per iteration)", 1);
    run<53>("unsynchronised: 16 mfma | 44 plain + 32 exp2 (per iteration)", 1);
    run<51>("ping-pong:      16 mfma | 44 plain + 32 exp2 (per iteration)", 1);
    run<55>("unsynchronised, vector phase at priority 2 (44 plain mix)", 1);
    run<54>("ping-pong,      vector phase at priority 2 (44 plain mix)", 1);
    run<57>("unsynchronised, MFMA phase at priority 2 (44 plain mix)", 1);
    run<56>("ping-pong,      MFMA phase at priority 2 (44 plain mix)", 1);
    return 0;
  }
  if (getenv("MFMA_SHAPES")) {      // round 4: 32x32x16 against 16x16x32 at equal FLOPs and equal vector work, per iteration
    run<12>("phase: 16 mfma 32x32x16 (per iteration)", 1);
    run<40>("phase: 32 mfma 16x16x32 (per iteration)", 1);
    run<10>("phases: 16 mfma32 | 96 plain + 32 exp2 (per iteration)", 1);
    run<41>("phases: 32 mfma16 | 96 plain + 32 exp2 (per iteration)", 1);
    run<43>("16 x (1 mfma32, 2 exp2, 6 plain) (per iteration)", 1);
    run<42>("16 x (2 mfma16, 2 exp2, 6 plain) (per iteration)", 1);
    return 0;
  }
  run<0>("v_exp_f32", 16); run<1>("v_add_f32", 16); run<5>("v_rcp_f32", 16); run<2>("v_exp + v_add (pairs)", 32);
  run<4>("mfma 32x32x16 alone", 4); run<3>("4 v_exp per mfma (count exp)", 16);
  run<9>("mfma, 2 accumulators (per mfma)", 8);
  run<6>("mfma + 6 plain + 2 exp2, 2 acc (per mfma)", 8);
  run<7>("mfma + 6 plain + 2 exp2, 1 acc (per mfma)", 8);
  run<8>("mfma + 8 plain, 2 acc (per mfma)", 8);
  run<20>("8 ds_read_b128 + 16 ds_read_b64 (per iteration)", 1);
  run<21>("64 s_add (per iteration)", 1);
  run<22>("4 global_load_lds_dwordx4 (per iteration)", 1);
  run<23>("s_barrier (per iteration)", 1);
  run<24>("8 x (v_cmp, untaken branch, v_mul) (per iteration)", 1);
  run<12>("phase: 16 mfma (per iteration)", 1);
  run<13>("phase: 96 plain + 32 exp2 (per iteration)", 1);
  run<11>("phases: 16 mfma | 96 plain (per iteration)", 1);
  run<10>("phases: 16 mfma | 96 plain + 32 exp2 (per iteration)", 1);
  run<30>("32 x (mfma, 2 exp2, 4 plain) (per mfma)", 32);
  run<32>("32 x (mfma, 2 exp2, 4 plain), 4 acc (per mfma)", 32);
  run<31>("32 x (mfma, 2 exp2, 2 plain) (per mfma)", 32);
  run<33>("32 x (mfma, 1 exp2, 2 plain) (per mfma)", 32);
  return 0;
}

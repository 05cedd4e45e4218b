// LDS read throughput per CU by instruction and address pattern (round 4): is the attention loop's fragment traffic - per 64-key
// tile and wave 8 ds_read_b128 (K, swizzled rows) + 16 ds_read_b64_tr_b16 (V^T) - bounded by the LDS pipe all four SIMDs share?
// Diagnostic; hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_rates.hip -o /tmp/lds_rates
//
// One block per CU of 4 w waves (w = 1..4 waves per SIMD).  Per iteration a wave issues all reads of its kind, waits once
// (lgkmcnt(0)) and folds the results into one register.  Printed: block makespan / (iters x w) = cycles per iteration and SIMD,
// and bytes per cycle and CU = 4 x bytes per wave-iteration / that.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define RD128(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:" #off_ : "=v"(dst_) : "v"(addr_))
#define RD64(dst_, addr_, off_) asm volatile("ds_read_b64 %0, %1 offset:" #off_ : "=v"(dst_) : "v"(addr_))
#define RD64T(dst_, addr_, off_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off_ : "=v"(dst_) : "v"(addr_))
#define RD32(dst_, addr_, off_) asm volatile("ds_read_b32 %0, %1 offset:" #off_ : "=v"(dst_) : "v"(addr_))

template <int KIND>
__global__ void k(unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  // the attention kernel's K fragment offsets (ttv_attn.hip: koff0..3) and V^T offsets (voff_d0 / voff_d1)
  const int ksw = (r >> 1) & 7;
  const unsigned ko0 = lds0 + r * 128 + (((0 * 2 + h) ^ ksw) << 4), ko1 = lds0 + r * 128 + (((1 * 2 + h) ^ ksw) << 4);
  const unsigned ko2 = lds0 + r * 128 + (((2 * 2 + h) ^ ksw) << 4), ko3 = lds0 + r * 128 + (((3 * 2 + h) ^ ksw) << 4);
  const int gi = lane & 15, tq = gi >> 2, tp = gi & 3, g16 = (lane >> 4) & 1;
  const int vsw = (tq >> 1) & 1;
  const int vlane = (4 * h + tq) * 128 + (g16 * 2 + (tp >> 1)) * 16 + (tp & 1) * 8;
  const unsigned vo0 = lds0 + 16384 + vlane + (vsw ? 64 : 0), vo1 = lds0 + 16384 + vlane + (vsw ? 0 : 64);
  const unsigned lin16 = lds0 + lane * 16, lin8 = lds0 + lane * 8, lin4 = lds0 + lane * 4;
  // the attention backward's 64 x 64 tiles (ttv_bwd.hip: 128-byte rows, chunk swizzle TSW(row)): frag_colp's transposed reads (16 lanes = 4
  // consecutive rows x 32 bytes; second read 16 rows on) and frag_row's b128 reads (16 lanes = 16 rows of one chunk), under three swizzles:
  //   SW 0: (r >> 1) & 7 (the kernel's)   SW 1: ((r >> 1) & 3) << 1 | (r >> 3) & 1   SW 2: pairs by (m & 1) | ((m >> 1 ^ m >> 2) & 1) << 1, m = (r >> 1) & 7
  constexpr int SW = (KIND >= 7) ? (KIND - 7) % 3 : 0;
  auto tsw = [](int r_) {
    const int m = (r_ >> 1) & 7;
    return SW == 0 ? m : SW == 1 ? (((m & 3) << 1) | (m >> 2)) : ((((m & 1) | ((((m >> 1) ^ (m >> 2)) & 1) << 1)) << 1) | (m >> 2));
  };
  const int bkq = lane >> 4, bl15 = lane & 15;
  unsigned cp[4], rw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = bkq * 4 + tq, cc = i * 16 + tp * 4;
    cp[i] = lds0 + rr * 128 + ((((cc >> 3) ^ tsw(rr)) << 4) | ((cc & 7) << 1));      // rows rr + 16 / + 32 / + 48 have the same swizzle value: plain offsets
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) rw[ks] = lds0 + bl15 * 128 + (((ks * 4 + bkq) ^ tsw(bl15)) << 4);        // rows + 16: same swizzle
  __syncthreads();
  unsigned vx = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {            // 16 ds_read_b128, lanes linear (16 KB per wave)
      u32x4 q[16];
      RD128(q[0], lin16, 0); RD128(q[1], lin16, 1024); RD128(q[2], lin16, 2048); RD128(q[3], lin16, 3072);
      RD128(q[4], lin16, 4096); RD128(q[5], lin16, 5120); RD128(q[6], lin16, 6144); RD128(q[7], lin16, 7168);
      RD128(q[8], lin16, 8192); RD128(q[9], lin16, 9216); RD128(q[10], lin16, 10240); RD128(q[11], lin16, 11264);
      RD128(q[12], lin16, 12288); RD128(q[13], lin16, 13312); RD128(q[14], lin16, 14336); RD128(q[15], lin16, 15360);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) vx ^= q[i].x ^ q[i].w;
    }
    if (KIND == 1) {            // 32 ds_read_b64, lanes linear (16 KB per wave)
      u32x2 q[32];
      RD64(q[0], lin8, 0); RD64(q[1], lin8, 512); RD64(q[2], lin8, 1024); RD64(q[3], lin8, 1536);
      RD64(q[4], lin8, 2048); RD64(q[5], lin8, 2560); RD64(q[6], lin8, 3072); RD64(q[7], lin8, 3584);
      RD64(q[8], lin8, 4096); RD64(q[9], lin8, 4608); RD64(q[10], lin8, 5120); RD64(q[11], lin8, 5632);
      RD64(q[12], lin8, 6144); RD64(q[13], lin8, 6656); RD64(q[14], lin8, 7168); RD64(q[15], lin8, 7680);
      RD64(q[16], lin8, 8192); RD64(q[17], lin8, 8704); RD64(q[18], lin8, 9216); RD64(q[19], lin8, 9728);
      RD64(q[20], lin8, 10240); RD64(q[21], lin8, 10752); RD64(q[22], lin8, 11264); RD64(q[23], lin8, 11776);
      RD64(q[24], lin8, 12288); RD64(q[25], lin8, 12800); RD64(q[26], lin8, 13312); RD64(q[27], lin8, 13824);
      RD64(q[28], lin8, 14336); RD64(q[29], lin8, 14848); RD64(q[30], lin8, 15360); RD64(q[31], lin8, 15872);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) vx ^= q[i].x ^ q[i].y;
    }
    if (KIND == 2) {            // 16 ds_read_b128 in the K fragment pattern (two tiles' worth: 16 KB per wave)
      u32x4 q[16];
      RD128(q[0], ko0, 0); RD128(q[1], ko1, 0); RD128(q[2], ko2, 0); RD128(q[3], ko3, 0);
      RD128(q[4], ko0, 4096); RD128(q[5], ko1, 4096); RD128(q[6], ko2, 4096); RD128(q[7], ko3, 4096);
      RD128(q[8], ko0, 8192); RD128(q[9], ko1, 8192); RD128(q[10], ko2, 8192); RD128(q[11], ko3, 8192);
      RD128(q[12], ko0, 12288); RD128(q[13], ko1, 12288); RD128(q[14], ko2, 12288); RD128(q[15], ko3, 12288);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) vx ^= q[i].x ^ q[i].w;
    }
    if (KIND == 3) {            // 32 ds_read_b64_tr_b16 in the V^T fragment pattern (two tiles' worth: 16 KB per wave)
      u32x2 q[32];
      RD64T(q[0], vo0, 0); RD64T(q[1], vo0, 1024); RD64T(q[2], vo0, 2048); RD64T(q[3], vo0, 3072);
      RD64T(q[4], vo0, 4096); RD64T(q[5], vo0, 5120); RD64T(q[6], vo0, 6144); RD64T(q[7], vo0, 7168);
      RD64T(q[8], vo1, 0); RD64T(q[9], vo1, 1024); RD64T(q[10], vo1, 2048); RD64T(q[11], vo1, 3072);
      RD64T(q[12], vo1, 4096); RD64T(q[13], vo1, 5120); RD64T(q[14], vo1, 6144); RD64T(q[15], vo1, 7168);
      RD64T(q[16], vo0, 8192); RD64T(q[17], vo0, 9216); RD64T(q[18], vo0, 10240); RD64T(q[19], vo0, 11264);
      RD64T(q[20], vo0, 12288); RD64T(q[21], vo0, 13312); RD64T(q[22], vo0, 14336); RD64T(q[23], vo0, 15360);
      RD64T(q[24], vo1, 8192); RD64T(q[25], vo1, 9216); RD64T(q[26], vo1, 10240); RD64T(q[27], vo1, 11264);
      RD64T(q[28], vo1, 12288); RD64T(q[29], vo1, 13312); RD64T(q[30], vo1, 14336); RD64T(q[31], vo1, 15360);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) vx ^= q[i].x ^ q[i].y;
    }
    if (KIND == 4) {            // one attention tile: 8 K reads + 16 V^T reads (16 KB per wave)
      u32x4 q[8];
      u32x2 p[16];
      RD128(q[0], ko0, 0); RD128(q[1], ko1, 0); RD128(q[2], ko2, 0); RD128(q[3], ko3, 0);
      RD128(q[4], ko0, 4096); RD128(q[5], ko1, 4096); RD128(q[6], ko2, 4096); RD128(q[7], ko3, 4096);
      RD64T(p[0], vo0, 0); RD64T(p[1], vo0, 1024); RD64T(p[2], vo0, 2048); RD64T(p[3], vo0, 3072);
      RD64T(p[4], vo0, 4096); RD64T(p[5], vo0, 5120); RD64T(p[6], vo0, 6144); RD64T(p[7], vo0, 7168);
      RD64T(p[8], vo1, 0); RD64T(p[9], vo1, 1024); RD64T(p[10], vo1, 2048); RD64T(p[11], vo1, 3072);
      RD64T(p[12], vo1, 4096); RD64T(p[13], vo1, 5120); RD64T(p[14], vo1, 6144); RD64T(p[15], vo1, 7168);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) vx ^= q[i].x ^ q[i].w;
#pragma unroll
      for (int i = 0; i < 16; ++i) vx ^= p[i].x ^ p[i].y;
    }
    if (KIND >= 7 && KIND <= 9) {   // 32 transposed reads of the attention backward (frag_colp on two tiles, both halves of the rows; 16 KB per wave)
      u32x2 q[32];
      RD64T(q[0], cp[0], 0); RD64T(q[1], cp[0], 2048); RD64T(q[2], cp[1], 0); RD64T(q[3], cp[1], 2048);
      RD64T(q[4], cp[2], 0); RD64T(q[5], cp[2], 2048); RD64T(q[6], cp[3], 0); RD64T(q[7], cp[3], 2048);
      RD64T(q[8], cp[0], 4096); RD64T(q[9], cp[0], 6144); RD64T(q[10], cp[1], 4096); RD64T(q[11], cp[1], 6144);
      RD64T(q[12], cp[2], 4096); RD64T(q[13], cp[2], 6144); RD64T(q[14], cp[3], 4096); RD64T(q[15], cp[3], 6144);
      RD64T(q[16], cp[0], 8192); RD64T(q[17], cp[0], 10240); RD64T(q[18], cp[1], 8192); RD64T(q[19], cp[1], 10240);
      RD64T(q[20], cp[2], 8192); RD64T(q[21], cp[2], 10240); RD64T(q[22], cp[3], 8192); RD64T(q[23], cp[3], 10240);
      RD64T(q[24], cp[0], 12288); RD64T(q[25], cp[0], 14336); RD64T(q[26], cp[1], 12288); RD64T(q[27], cp[1], 14336);
      RD64T(q[28], cp[2], 12288); RD64T(q[29], cp[2], 14336); RD64T(q[30], cp[3], 12288); RD64T(q[31], cp[3], 14336);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) vx ^= q[i].x ^ q[i].y;
    }
    if (KIND >= 10 && KIND <= 12) { // 16 b128 row-fragment reads of the attention backward (frag_row: two k-steps x 16-row groups; 16 KB per wave)
      u32x4 q[16];
      RD128(q[0], rw[0], 0); RD128(q[1], rw[1], 0); RD128(q[2], rw[0], 2048); RD128(q[3], rw[1], 2048);
      RD128(q[4], rw[0], 4096); RD128(q[5], rw[1], 4096); RD128(q[6], rw[0], 6144); RD128(q[7], rw[1], 6144);
      RD128(q[8], rw[0], 8192); RD128(q[9], rw[1], 8192); RD128(q[10], rw[0], 10240); RD128(q[11], rw[1], 10240);
      RD128(q[12], rw[0], 12288); RD128(q[13], rw[1], 12288); RD128(q[14], rw[0], 14336); RD128(q[15], rw[1], 14336);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) vx ^= q[i].x ^ q[i].w;
    }
    if (KIND == 5) {            // 32 ds_read_b32, lanes linear (8 KB per wave)
      unsigned q[32];
      RD32(q[0], lin4, 0); RD32(q[1], lin4, 256); RD32(q[2], lin4, 512); RD32(q[3], lin4, 768);
      RD32(q[4], lin4, 1024); RD32(q[5], lin4, 1280); RD32(q[6], lin4, 1536); RD32(q[7], lin4, 1792);
      RD32(q[8], lin4, 2048); RD32(q[9], lin4, 2304); RD32(q[10], lin4, 2560); RD32(q[11], lin4, 2816);
      RD32(q[12], lin4, 3072); RD32(q[13], lin4, 3328); RD32(q[14], lin4, 3584); RD32(q[15], lin4, 3840);
      RD32(q[16], lin4, 4096); RD32(q[17], lin4, 4352); RD32(q[18], lin4, 4608); RD32(q[19], lin4, 4864);
      RD32(q[20], lin4, 5120); RD32(q[21], lin4, 5376); RD32(q[22], lin4, 5632); RD32(q[23], lin4, 5888);
      RD32(q[24], lin4, 6144); RD32(q[25], lin4, 6400); RD32(q[26], lin4, 6656); RD32(q[27], lin4, 6912);
      RD32(q[28], lin4, 7168); RD32(q[29], lin4, 7424); RD32(q[30], lin4, 7680); RD32(q[31], lin4, 7936);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) vx ^= q[i];
    }
    if (KIND == 6) {            // the K fragment bytes as 32 ds_read_b64 (each 16-byte chunk in two halves; 16 KB per wave)
      u32x2 q[32];
      RD64(q[0], ko0, 0); RD64(q[1], ko0, 8); RD64(q[2], ko1, 0); RD64(q[3], ko1, 8);
      RD64(q[4], ko2, 0); RD64(q[5], ko2, 8); RD64(q[6], ko3, 0); RD64(q[7], ko3, 8);
      RD64(q[8], ko0, 4096); RD64(q[9], ko0, 4104); RD64(q[10], ko1, 4096); RD64(q[11], ko1, 4104);
      RD64(q[12], ko2, 4096); RD64(q[13], ko2, 4104); RD64(q[14], ko3, 4096); RD64(q[15], ko3, 4104);
      RD64(q[16], ko0, 8192); RD64(q[17], ko0, 8200); RD64(q[18], ko1, 8192); RD64(q[19], ko1, 8200);
      RD64(q[20], ko2, 8192); RD64(q[21], ko2, 8200); RD64(q[22], ko3, 8192); RD64(q[23], ko3, 8200);
      RD64(q[24], ko0, 12288); RD64(q[25], ko0, 12296); RD64(q[26], ko1, 12288); RD64(q[27], ko1, 12296);
      RD64(q[28], ko2, 12288); RD64(q[29], ko2, 12296); RD64(q[30], ko3, 12288); RD64(q[31], ko3, 12296);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) vx ^= q[i].x ^ q[i].y;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = vx;
  if (threadIdx.x % 64 == 0) {
    cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t0;
    cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = t1;
  }
}

template <int KIND>
void run(const char* name, int bytes_per_wave_iter) {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 2 * 8192 * 8);
  for (int wps = 1; wps <= 4; ++wps) {
    const int threads = 256 * wps, iters = 200;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    const int wpb = threads / 64;
    std::vector<long long> h(2 * 256 * wpb);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> span(256);
    for (int b = 0; b < 256; ++b) {
      long long lo = h[2 * b * wpb], hi = h[2 * b * wpb + 1];
      for (int w = 0; w < wpb; ++w) {
        const long long a0 = h[2 * (b * wpb + w)], a1 = h[2 * (b * wpb + w) + 1];
        lo = a0 < lo ? a0 : lo; hi = a1 > hi ? a1 : hi;
      }
      span[b] = (double)(hi - lo);
    }
    std::sort(span.begin(), span.end());
    const double per = span[128] / ((double)iters * wps);
    printf("%-64s %d waves/SIMD: %7.1f cycles per iteration and SIMD = %6.1f bytes / cycle / CU\n", name, wps, per, 4.0 * bytes_per_wave_iter / per);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("16 ds_read_b128, lanes linear", 16384);
  run<1>("32 ds_read_b64, lanes linear", 16384);
  run<5>("32 ds_read_b32, lanes linear", 8192);
  run<2>("16 ds_read_b128, attention K fragment pattern (swizzled rows)", 16384);
  run<6>("the same bytes as 32 ds_read_b64", 16384);
  run<3>("32 ds_read_b64_tr_b16, attention V^T fragment pattern", 16384);
  run<4>("one attention tile: 8 K reads + 16 V^T reads", 16384);
  run<7>("attention backward: 32 transposed reads, swizzle (r>>1)&7", 16384);
  run<8>("attention backward: 32 transposed reads, swizzle pairs by (r>>1)&3", 16384);
  run<9>("attention backward: 32 transposed reads, swizzle pairs mixed", 16384);
  run<10>("attention backward: 16 b128 row reads, swizzle (r>>1)&7", 16384);
  run<11>("attention backward: 16 b128 row reads, swizzle pairs by (r>>1)&3", 16384);
  run<12>("attention backward: 16 b128 row reads, swizzle pairs mixed", 16384);
  return 0;
}

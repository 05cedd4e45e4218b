// LDS-DMA (global_load_lds_dwordx4) streaming rate per CU from an L2-resident / MALL-resident source, by waves per CU and by the number
// of 1 KB instructions each wave keeps in flight.  One block per CU; every wave walks its own share of the source in 1 KB pieces into
// its own LDS ring.  Diagnostic; hipcc --offload-arch=gfx950 tools/ubench/dma_rates.hip -o tools/ubench/dma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int DEPTH>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ src, size_t src_bytes, int pieces_per_wave, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + wave * (DEPTH * 1024);
  // piece p of this wave: bytes [(p * nw + wave) * 1024 .. + 1023] of the source, wrapped; every CU streams the same bytes (weights)
  const uint32_t voff = lane * 16;
  size_t pos = (size_t)wave * 1024;
  const size_t step = (size_t)nw * 1024;
#pragma unroll 1
  for (int p = 0; p < pieces_per_wave; ++p) {
    const char* base = src + pos;
    pos += step;
    if (pos >= src_bytes) pos -= src_bytes;
    const uint32_t dst = lds0 + (p % DEPTH) * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\t"
                 "s_waitcnt vmcnt(%4)"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst), "n"(DEPTH - 1) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = *reinterpret_cast<float*>(lds);
}

template <int DEPTH>
static void run(const char* name, const char* src, size_t src_bytes, int waves, float* sink) {
  const int pieces = 4096 / waves * 4;                       // 16 MB per CU in all
  const size_t lds = (size_t)waves * DEPTH * 1024;
  if (lds > 160 * 1024) return;
  hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(waves * 64), lds, 0, src, src_bytes, pieces, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(waves * 64), lds, 0, src, src_bytes, pieces, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes_cu = (double)pieces * waves * 1024;
  printf("%-14s waves/CU %2d  in flight per wave %2d (%3d KB per CU): %6.1f GB/s per CU  %6.2f TB/s chip\n", name, waves, DEPTH, waves * DEPTH,
         bytes_cu / (ms / 5 * 1e-3) / 1e9, 256 * bytes_cu / (ms / 5 * 1e-3) / 1e12);
}

int main() {
  char* big; float* sink;
  hipMalloc(&big, (size_t)512 << 20);
  hipMemset(big, 1, (size_t)512 << 20);
  hipMalloc(&sink, 256 * 4);
  struct { const char* name; size_t bytes; } srcs[] = {{"384 KB (L2)", 384 << 10}, {"1 MB (L2)", 1 << 20}, {"64 MB (MALL)", (size_t)64 << 20}, {"512 MB (HBM)", (size_t)512 << 20}};
  for (auto& s : srcs)
    for (int waves : {4, 8, 16}) {
      run<1>(s.name, big, s.bytes, waves, sink);
      run<2>(s.name, big, s.bytes, waves, sink);
      run<4>(s.name, big, s.bytes, waves, sink);
      run<8>(s.name, big, s.bytes, waves, sink);
      run<16>(s.name, big, s.bytes, waves, sink);
    }
  return 0;
}

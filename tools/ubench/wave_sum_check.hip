// Bit-identity of the VALU-only wave_sum / wave_max (ttv_common.h) against the ds_bpermute butterfly they replace, on random data over 40 binades.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ititok_video_amd/csrc -Iinclude tools/ubench/wave_sum_check.hip -o tools/ubench/wave_sum_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ttv_common.h"
__device__ __forceinline__ float ref_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float ref_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__global__ void k(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float v = in[i];
  out[i] = wave_sum(v); out[n + i] = ref_sum(v); out[2 * n + i] = wave_max(v); out[3 * n + i] = ref_max(v);
  out[4 * n + i] = wave_xor_dpp8(v); out[5 * n + i] = __shfl_xor(v, 8, 64); out[6 * n + i] = wave_xor_dpp4(v); out[7 * n + i] = __shfl_xor(v, 4, 64);
}
int main() {
  const int n = 256 * 64;
  float* h = (float*)malloc(n * 4);
  srand(3);
  for (int i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * expf((rand() % 40) - 20.f);
  float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 8 * n * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
  float* r = (float*)malloc(8 * n * 4);
  hipMemcpy(r, o, 8 * n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int p = 0; p < 4; ++p) bad += memcmp(r + 2 * p * n, r + (2 * p + 1) * n, n * 4) != 0;
  printf("wave_sum / wave_max / xor8 / xor4 against the ds_bpermute butterfly: %s (%d of 4 differ)\n", bad ? "MISMATCH" : "bit-identical", bad);
  return bad;
}

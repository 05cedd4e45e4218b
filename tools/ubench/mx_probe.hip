// Probe of v_mfma_scale_f32_16x16x128_f8f6f4's scale-operand semantics (which lane's scale byte multiplies which operand elements).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
// A = B = 1.0 (e4m3 0x38) everywhere, scales 2^0 except ONE lane's A (or B) scale byte = 2^1.  D[i][j] = 128 + (elements scaled).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OPSEL>
__global__ void probe(const int* sa, const int* sb, const unsigned char* adata, const unsigned char* bdata, float* out) {
  const int lane = threadIdx.x;
  v8i32 a, b;
  const int* ap = reinterpret_cast<const int*>(adata + lane * 32);
  const int* bp = reinterpret_cast<const int*>(bdata + lane * 32);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPSEL, sa[lane], OPSEL, sb[lane]);
  for (int r = 0; r < 4; ++r) out[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];     // D[row = 4 (lane / 16) + r][col = lane % 16]
}

int main() {
  int *sa, *sb; unsigned char *ad, *bd; float* out;
  hipMallocManaged(&sa, 256); hipMallocManaged(&sb, 256); hipMallocManaged(&ad, 2048); hipMallocManaged(&bd, 2048); hipMallocManaged(&out, 1024);
  // 1. scale mapping, byte 0, opsel 0
  for (int which = 0; which < 2; ++which) {
    printf("== one lane's %s scale = 2^1 (byte 0, opsel 0): rows / cols whose D changed, and by how much\n", which ? "B" : "A");
    for (int L = 0; L < 64; ++L) {
      for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
      (which ? sb : sa)[L] = 0x7F7F7F80;
      for (int i = 0; i < 2048; ++i) { ad[i] = 0x38; bd[i] = 0x38; }
      probe<0><<<1, 64>>>(sa, sb, ad, bd, out);
      hipDeviceSynchronize();
      printf("lane %2d:", L);
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (out[i * 16 + j] != 128.f && (which ? i == 0 : j == 0)) printf(" %s%d:+%g", which ? "col" : "row", which ? j : i, out[i * 16 + j] - 128.f);
      printf("\n");
    }
  }
  // 2. opsel: scale byte placed in byte p, opsel p
  printf("== opsel: lane 5's A scale 2^1 in byte p, others 2^0; D[row 5][0] - 128 with opsel 0..3\n");
  for (int p = 0; p < 4; ++p) {
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
    sa[5] = 0x7F7F7F7F + (1 << (8 * p));
    float r[4];
    probe<0><<<1, 64>>>(sa, sb, ad, bd, out); hipDeviceSynchronize(); r[0] = out[5 * 16];
    probe<1><<<1, 64>>>(sa, sb, ad, bd, out); hipDeviceSynchronize(); r[1] = out[5 * 16];
    probe<2><<<1, 64>>>(sa, sb, ad, bd, out); hipDeviceSynchronize(); r[2] = out[5 * 16];
    probe<3><<<1, 64>>>(sa, sb, ad, bd, out); hipDeviceSynchronize(); r[3] = out[5 * 16];
    printf("byte %d: opsel0 %+g opsel1 %+g opsel2 %+g opsel3 %+g\n", p, r[0] - 128, r[1] - 128, r[2] - 128, r[3] - 128);
  }
  // 3. data mapping: A one-hot byte (lane La, byte ba) = 1.0, rest 0; B = 1.0 in ONE (lane, byte), rest 0: which B (lane, byte) pairs with A's
  printf("== K pairing: A element (lane 16*g + 3, byte ba) pairs with B element (lane 16*g' + 7, byte bb)\n");
  for (int g = 0; g < 4; ++g) for (int ba = 0; ba < 32; ba += 5) {
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
    for (int i = 0; i < 2048; ++i) { ad[i] = 0; bd[i] = 0; }
    ad[(16 * g + 3) * 32 + ba] = 0x38;
    int found = 0;
    for (int g2 = 0; g2 < 4 && !found; ++g2) for (int bb = 0; bb < 32 && !found; ++bb) {
      for (int i = 0; i < 2048; ++i) bd[i] = 0;
      bd[(16 * g2 + 7) * 32 + bb] = 0x38;
      probe<0><<<1, 64>>>(sa, sb, ad, bd, out); hipDeviceSynchronize();
      if (out[3 * 16 + 7] == 1.f) { printf("A(g %d, byte %2d) <-> B(g %d, byte %2d)\n", g, ba, g2, bb); found = 1; }
    }
    if (!found) printf("A(g %d, byte %2d): no partner found\n", g, ba);
  }
  // 4. which lane's A scale multiplies the A element held at (lane 16 g + 3, byte ba)?  (and the same for B)
  printf("== scale ownership: element at (group g, byte b) of row 3 is multiplied by the scale byte of lane ...\n");
  for (int which = 0; which < 2; ++which)
    for (int g = 0; g < 4; ++g) for (int ba = 0; ba < 32; ba += 3) {
      for (int i = 0; i < 2048; ++i) { ad[i] = which ? 0x38 : 0; bd[i] = which ? 0 : 0x38; }
      (which ? bd : ad)[(16 * g + 3) * 32 + ba] = 0x38;
      if (which) for (int i = 0; i < 2048; ++i) ad[i] = 0x38;
      printf("%s(g %d, byte %2d):", which ? "B" : "A", g, ba);
      for (int gs = 0; gs < 4; ++gs) {
        for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
        (which ? sb : sa)[16 * gs + 3] = 0x7F7F7F80;
        probe<0><<<1, 64>>>(sa, sb, ad, bd, out); hipDeviceSynchronize();
        const float v = which ? out[0 * 16 + 3] : out[3 * 16 + 0];
        if (v == 2.f) printf(" lane group %d", gs);
      }
      printf("\n");
    }
  return 0;
}

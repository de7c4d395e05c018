// Probe (round 6): is v_mfma_f32_16x16x4_f32 one fp32 FMA chain per output element in k order (k = lane / 16 inside an
// instruction), like v_mfma_f32_32x32x2_f32 (k = lane / 32)?  If so a 16-row tile path can reproduce the layer-wise
// launches' contraction order bit for bit.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma16_probe.hip -o tools/bin/mfma16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k16(const float* A, const float* B, float* C, int K) {   // A [16][K], B [16][K], C [16][16]
  const int lane = threadIdx.x, c16 = lane & 15, g4 = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < K / 4; ++j)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c16 * K + 4 * j + g4], B[c16 * K + 4 * j + g4], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * g4 + r) * 16 + c16] = acc[r];
}
__global__ void k32(const float* A, const float* B, float* C, int K) {   // rows / cols 16..31 zero
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j = 0; j < K / 2; ++j) {
    const float a = l31 < 16 ? A[l31 * K + 2 * j + h] : 0.f, b = l31 < 16 ? B[l31 * K + 2 * j + h] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < 16 && l31 < 16) C[row * 16 + l31] = acc[r];
  }
}

int main() {
  const int K = 256;
  std::vector<float> A(16 * K), B(16 * K), C16(256), C32(256);
  srand(7);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k16<<<1, 64>>>(dA, dB, dC, K);
  hipMemcpy(C16.data(), dC, 1024, hipMemcpyDeviceToHost);
  k32<<<1, 64>>>(dA, dB, dC, K);
  hipMemcpy(C32.data(), dC, 1024, hipMemcpyDeviceToHost);
  int eq_chain16 = 0, eq_chain32 = 0, eq_16_32 = 0, eq_pair16 = 0;
  double maxd = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float chain = 0.f, pair = 0.f;
      for (int k = 0; k < K; ++k) chain = fmaf(A[i * K + k], B[j * K + k], chain);
      for (int k = 0; k < K; k += 4) {      // alternative: products summed pairwise, then added
        float p0 = A[i * K + k] * B[j * K + k], p1 = A[i * K + k + 1] * B[j * K + k + 1];
        float p2 = A[i * K + k + 2] * B[j * K + k + 2], p3 = A[i * K + k + 3] * B[j * K + k + 3];
        pair = pair + ((p0 + p1) + (p2 + p3));
      }
      eq_chain16 += C16[i * 16 + j] == chain, eq_chain32 += C32[i * 16 + j] == chain;
      eq_16_32 += C16[i * 16 + j] == C32[i * 16 + j], eq_pair16 += C16[i * 16 + j] == pair;
      maxd = fmax(maxd, fabs((double)C16[i * 16 + j] - (double)C32[i * 16 + j]));
    }
  printf("of 256 outputs: 16x16x4 == fmaf chain: %d   32x32x2 == fmaf chain: %d   16x16x4 == 32x32x2: %d   16x16x4 == pairwise: %d   max |16 - 32| = %.3g\n",
         eq_chain16, eq_chain32, eq_16_32, eq_pair16, maxd);
  return 0;
}

// Third probe: does the accumulator register class matter?  Same loop (8 back-to-back v_mfma_f32_32x32x2_f32, one
// accumulator, 512 threads, 1 workgroup per CU) with the accumulator forced into ArchVGPRs ("+v") or AccVGPRs ("+a").
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int AGPR>
__global__ __launch_bounds__(512) void probe(float* out, const float* inp, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float a = inp[threadIdx.x], b = a + 1.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
    a += 1e-9f;
  }
  float s = 0.0f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int AGPR>
void run(const char* what) {
  const int blocks = 256, iters = 4000;
  float *out, *inp;
  hipMalloc(&out, sizeof(float) * blocks * 512);
  hipMalloc(&inp, sizeof(float) * 512);
  hipMemset(inp, 0, sizeof(float) * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<AGPR><<<blocks, 512>>>(out, inp, 10);
  hipEventRecord(e0);
  probe<AGPR><<<blocks, 512>>>(out, inp, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s : %.1f ns per MFMA per SIMD\n", what, ms * 1e6 / (2.0 * iters * 8.0));
}

int main() {
  run<1>("accumulator in AccVGPRs (a[..])");
  run<0>("accumulator in ArchVGPRs (v[..])");
  return 0;
}

// Issue-rate probe for v_mfma_f32_32x32x2_f32 on gfx950: cycles per MFMA and SIMD as a function of the number of waves per
// SIMD and of independent accumulators per wave.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-3f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int threads, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<NACC><<<blocks, threads>>>(out, cyc, 10);
  hipEventRecord(e0);
  probe<NACC><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const double waves_per_simd = threads / 64.0 * blocks_per_cu / 4.0;
  const double mfma_per_simd = waves_per_simd * iters * 16.0;
  // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: convert through the event time instead
  printf("acc/wave %d  threads %4d  blocks/CU %d  waves/SIMD %.0f : %.1f ns per MFMA per SIMD  (%.1f TFLOP/s)\n", NACC, threads,
         blocks_per_cu, waves_per_simd, ms * 1e6 / mfma_per_simd, 256 * 4 * mfma_per_simd * 4096.0 / (ms * 1e-3) / 1e12);
  hipFree(out), hipFree(cyc);
}

int main() {
  run<1>(256, 1), run<2>(256, 1), run<4>(256, 1);
  run<1>(512, 1), run<2>(512, 1), run<4>(512, 1);
  run<1>(256, 2), run<2>(256, 2), run<4>(256, 2);
  run<1>(1024, 1), run<4>(1024, 1);
  return 0;
}

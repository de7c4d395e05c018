// Second issue-rate probe: the fused_fwd_kernel shape (512 threads, 1 workgroup per CU, big dynamic LDS, MFMA blocks of 8
// with different operand registers, a few VALU between blocks).  hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(float* out, const float* inp, int iters) {
  extern __shared__ float smem[];
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float4 a0 = *reinterpret_cast<const float4*>(inp + threadIdx.x * 4), a1 = a0, b0 = a0, b1 = a0;
  a1.x += 1, b0.y += 2, b1.z += 3;
  int ap = threadIdx.x;
  if (VARIANT == 2) smem[threadIdx.x] = a0.x;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (VARIANT >= 1) {          // a few VALU between the blocks, like the fused loop's address update / fragment moves
      ap += 16;
      a0.x += 1e-9f * ap, b0.x -= 1e-9f * ap;
    }
    if (VARIANT == 2) {          // and four LDS reads whose results feed the next block
      const float4 t0 = *reinterpret_cast<const float4*>(smem + ((ap * 4) & 8188));
      const float4 t1 = *reinterpret_cast<const float4*>(smem + ((ap * 4 + 512) & 8188));
      a0.y = t0.x, a1.y = t0.y, b0.z = t1.x, b1.w = t1.y;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.0f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VARIANT>
void run(size_t lds, const char* what) {
  const int blocks = 256, iters = 4000;
  float *out, *inp;
  hipMalloc(&out, sizeof(float) * blocks * 512);
  hipMalloc(&inp, sizeof(float) * 4 * 512);
  hipMemset(inp, 0, sizeof(float) * 4 * 512);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<VARIANT><<<blocks, 512, lds>>>(out, inp, 10);
  hipEventRecord(e0);
  probe<VARIANT><<<blocks, 512, lds>>>(out, inp, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = 2.0 * iters * 8.0;      // two waves per SIMD
  printf("%-44s lds %6zu B : %.1f ns per MFMA per SIMD\n", what, lds, ms * 1e6 / mfma_per_simd);
  hipFree(out), hipFree(inp);
}

int main() {
  run<0>(32768, "8 back-to-back MFMAs, one accumulator");
  run<0>(131072, "same, 128 KB of LDS per workgroup");
  run<1>(131072, "+ VALU between the blocks");
  run<2>(131072, "+ 2 dependent ds_read_b128 per block");
  return 0;
}

// Probe (round 6): issue cost of v_permlane32_swap / v_permlane16_swap / ds_write_b128 / ds_read_b128 with 8 waves on one CU.
//   hipcc --offload-arch=gfx950 -O2 tools/swap_probe.hip -o tools/bin/swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f4v = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float* sink, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 32 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a = threadIdx.x, b = threadIdx.x * 2.f, c = 1.f, d = 3.f;
  float* wr = lds + wave * 32 * 36 + (lane >> 3) * 36 + 4 * (lane & 7);
  f4v v = {a, b, c, d};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
        c = __uint_as_float(q[0]), d = __uint_as_float(q[1]);
      }
    } else if (mode == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
        auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(d), false, false);
        c = __uint_as_float(q[0]), d = __uint_as_float(q[1]);
      }
    } else if (mode == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        *reinterpret_cast<f4v*>(wr + 8 * (k & 3) * 36) = v;
        asm volatile("" ::: "memory");
      }
    } else if (mode == 3) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        f4v r = *reinterpret_cast<const f4v*>(wr + 8 * (k & 3) * 36);
        asm volatile("" ::"v"(r) : "memory");
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) { a = a * 1.0001f + b; c = c * 1.0001f + d; }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  sink[threadIdx.x] = a + b + c + d + v.x;
}
int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 4096);
  const char* names[] = {"v_permlane32_swap", "v_permlane16_swap", "ds_write_b128", "ds_read_b128", "v_fma (2 per)"};
  const int per_iter[] = {16, 16, 16, 16, 32};
  for (int mode = 0; mode < 5; ++mode) {
    probe<<<1, 512>>>(out, sink, mode);
    probe<<<1, 512>>>(out, sink, mode);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%-20s 8 waves on one CU: %.1f shader-clock ticks per instruction and wave (wave 0), %.1f (wave 7)\n", names[mode],
           (double)h[0] / (256.0 * per_iter[mode]), (double)h[7] / (256.0 * per_iter[mode]));
  }
  return 0;
}

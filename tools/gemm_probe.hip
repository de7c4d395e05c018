// Standalone probe for the fp32-MFMA GEMM kernel (csrc/gemm_f32.h): times the three GEMM forms
// at the layer shapes of the hot path.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   tools/gemm_probe.hip -o tools/gemm_probe && tools/gemm_probe [reps]
#include "../constraints-as-terminations_amd/csrc/gemm_f32.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using gemm::Params;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

static size_t g_extra_lds = 0;   // pad the dynamic LDS request to force fewer workgroups per CU
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int BKT = 16>
float run(const Params& p, int reps) {
  dim3 grid(((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM), 1, p.nets * p.splits);
  const size_t lds = gemm::smem_bytes<BM, BN, A_KC, B_KC, BKT>() + g_extra_lds;
  if (lds > 64 * 1024)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, BKT>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, BKT><<<grid, 256, lds>>>(p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) gemm::gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, BKT><<<grid, 256, lds>>>(p);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int M = 16384;
  const size_t big = (size_t)M * 512;
  float *X[2], *Y[2], *W[2], *bias[2], *aux[2], *part, *dbp;
  for (int n = 0; n < 2; ++n) {
    CK(hipMalloc(&X[n], big * 4));
    CK(hipMalloc(&Y[n], big * 4));
    CK(hipMalloc(&aux[n], big * 4));
    CK(hipMalloc(&W[n], 512 * 512 * 4));
    CK(hipMalloc(&bias[n], 4096));
    std::vector<float> h(big);
    for (size_t i = 0; i < big; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    CK(hipMemcpy(X[n], h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Y[n], h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(aux[n], h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W[n], h.data(), 512 * 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias[n], 0, 4096));
  }
  CK(hipMalloc(&part, (size_t)2 * 64 * 512 * 512 * 4));
  CK(hipMalloc(&dbp, (size_t)2 * 64 * 4096 * 4));

  struct Shape { const char* name; int J, K; };
  const Shape fwd[] = {{"fwd  J=256 K=48 ", 256, 48}, {"fwd  J=256 K=256", 256, 256}, {"fwd  J=512 K=48 ", 512, 48},
                       {"fwd  J=256 K=512", 256, 512}, {"fwd  J=128 K=256", 128, 256}};
  for (const Shape& s : fwd) {
    Params p{};
    p.nets = 2, p.splits = 1, p.I = M, p.J = s.J, p.Kc = s.K, p.lda = s.K, p.ldb = s.K, p.ldc = s.J;
    for (int n = 0; n < 2; ++n) p.op[n] = {X[n], W[n], Y[n], bias[n], nullptr, nullptr};
    const double fl = 2.0 * M * s.J * s.K * 2;
    float a = run<128, 128, true, true, gemm::EPI_BIAS_ELU>(p, reps);
    float b = run<64, 64, true, true, gemm::EPI_BIAS_ELU>(p, reps);
    printf("%s  128x128: %7.1f us %6.1f TF   64x64: %7.1f us %6.1f TF", s.name, a, fl / a / 1e6, b, fl / b / 1e6);
    if (s.K % 32 == 0) {
      float c = run<128, 128, true, true, gemm::EPI_BIAS_ELU, 32>(p, reps);
      float d = run<64, 64, true, true, gemm::EPI_BIAS_ELU, 32>(p, reps);
      printf("   BK32 128x128: %7.1f us %6.1f TF  64x64: %7.1f us %6.1f TF", c, fl / c / 1e6, d, fl / d / 1e6);
    }
    printf("\n");
  }
  const Shape dx[] = {{"dx   J=256 Kc=256", 256, 256}, {"dx   J=512 Kc=256", 512, 256}, {"dx   J=256 Kc=128", 256, 128}};
  for (const Shape& s : dx) {
    Params p{};
    p.nets = 2, p.splits = 1, p.I = M, p.J = s.J, p.Kc = s.K, p.lda = s.K, p.ldb = s.J, p.ldc = s.J, p.ldaux = s.J;
    for (int n = 0; n < 2; ++n) p.op[n] = {X[n], W[n], Y[n], nullptr, aux[n], nullptr};
    const double fl = 2.0 * M * s.J * s.K * 2;
    float a = run<128, 128, true, false, gemm::EPI_MUL_DELU>(p, reps);
    float b = run<64, 64, true, false, gemm::EPI_MUL_DELU>(p, reps);
    float c = run<128, 128, true, false, gemm::EPI_MUL_DELU, 32>(p, reps);
    float d = run<64, 64, true, false, gemm::EPI_MUL_DELU, 32>(p, reps);
    printf("%s 128x128: %7.1f us %6.1f TF   64x64: %7.1f us %6.1f TF   BK32 128x128: %7.1f us %6.1f TF  64x64: %7.1f us %6.1f TF\n",
           s.name, a, fl / a / 1e6, b, fl / b / 1e6, c, fl / c / 1e6, d, fl / d / 1e6);
  }
  struct DW { const char* name; int I, J, splits; };
  const DW dw[] = {{"dw   256x256 s32", 256, 256, 32}, {"dw   256x256 s16", 256, 256, 16}, {"dw   256x512 s32", 256, 512, 32},
                   {"dw   256x48  s64", 256, 48, 64}, {"dw   128x256 s64", 128, 256, 64}};
  for (const DW& s : dw) {
    Params p{};
    p.nets = 2, p.splits = s.splits, p.I = s.I, p.J = s.J, p.Kc = M, p.lda = s.I, p.ldb = s.J, p.ldc = s.J;
    p.kc_per_split = M / s.splits, p.c_split_stride = 2 * (int64_t)s.I * s.J;
    for (int n = 0; n < 2; ++n) p.op[n] = {X[n], Y[n], part + (size_t)n * s.I * s.J, nullptr, nullptr, dbp + (size_t)n * s.splits * s.I};
    const double fl = 2.0 * M * s.I * s.J * 2;
    float a = run<128, 128, false, false, gemm::EPI_PARTIAL>(p, reps);
    float b = run<64, 64, false, false, gemm::EPI_PARTIAL>(p, reps);
    float c = run<128, 128, false, false, gemm::EPI_PARTIAL, 32>(p, reps);
    float d = run<64, 64, false, false, gemm::EPI_PARTIAL, 32>(p, reps);
    printf("%s  128x128: %7.1f us %6.1f TF   64x64: %7.1f us %6.1f TF   BK32 128x128: %7.1f us %6.1f TF  64x64: %7.1f us %6.1f TF\n",
           s.name, a, fl / a / 1e6, b, fl / b / 1e6, c, fl / c / 1e6, d, fl / d / 1e6);
  }
  // occupancy experiment: the same 128x128 forward GEMM with 1 workgroup (4 waves) per CU
  for (size_t extra : {(size_t)0, (size_t)60 * 1024, (size_t)110 * 1024}) {
    g_extra_lds = extra;
    Params p{};
    p.nets = 2, p.splits = 1, p.I = M, p.J = 256, p.Kc = 512, p.lda = 512, p.ldb = 512, p.ldc = 256;
    for (int n = 0; n < 2; ++n) p.op[n] = {X[n], W[n], Y[n], bias[n], nullptr, nullptr};
    const double fl = 2.0 * M * 256 * 512 * 2;
    float a = run<128, 128, true, true, gemm::EPI_BIAS_ELU>(p, reps);
    printf("occupancy probe fwd J=256 K=512 128x128, extra LDS %3zu KB: %7.1f us %6.1f TF\n", extra / 1024, a, fl / a / 1e6);
  }
  g_extra_lds = 0;
  return 0;
}

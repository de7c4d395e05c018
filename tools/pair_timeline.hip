// Timeline probe for the paired weight-gradient + data-gradient launch (csrc/gemm_f32.h gemm_pair_kernel) at the
// cfg2 layer shape (M = 16384, 256x256, both nets): records, per workgroup, the CU it ran on and its start / end
// clocks, for several orderings / split counts / tiles per workgroup / register caps, plus the empirical fp32-MFMA
// ceiling and shader clock.  Build here, run on the GPU box (the binary travels with the snapshot):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/pair_timeline.hip -o tools/bin/pair_timeline
// Findings of round 2 are in DESIGN.md section 4 ("Where the paired launch spends its time").
#define GEMM_TIMELINE
#include "../constraints-as-terminations_amd/csrc/gemm_f32.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
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

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

// order: 0 = problem 0 first (production), 1 = problem 0 last, 2 = one problem-0 workgroup every `stride` blocks
template <int BM0, int BN0, int BM1, int BN1, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void pair_tl(const Params p0, const Params p1, const int tiles0, const int n0,
                                               const int tiles1, const int n1, const int order, const int stride,
                                               Rec* rec, const int tpb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  unsigned long long t0 = 0;
  if (threadIdx.x == 0) t0 = wall_clock64();
  int kind, idx;
  if (order == 0) {
    kind = b < n0 ? 0 : 1, idx = b < n0 ? b : b - n0;
  } else if (order == 1) {
    kind = b < n1 ? 1 : 0, idx = b < n1 ? b : b - n1;
  } else {
    // blocks b with b % stride == 0 (while problem-0 items remain) run problem 0
    const int q = b / stride, r = b % stride;
    const int full = n0 * stride;        // first `full` blocks hold all n0 problem-0 items
    if (b < full) {
      if (r == 0) kind = 0, idx = q;
      else kind = 1, idx = q * (stride - 1) + (r - 1);
    } else {
      kind = 1, idx = n0 * (stride - 1) + (b - full);
    }
  }
  if (kind == 0) {
    gemm::gemm_body<BM0, BN0, false, false, gemm::EPI_PARTIAL>(p0, gemm::xcd_tile_index(idx % tiles0, tiles0), idx / tiles0, smem);
  } else if (tpb == 1) {
    gemm::gemm_body<BM1, BN1, true, false, gemm::EPI_MUL_DELU>(p1, gemm::xcd_tile_index(idx % tiles1, tiles1), idx / tiles1, smem);
  } else {
    // tpb consecutive tiles of this XCD's chunk per workgroup (n1 counts workgroups; both nets in one index space)
    for (int t = 0; t < tpb; ++t) {
      const int raw = (idx & 7) + 8 * (tpb * (idx >> 3) + t);      // same XCD residue, consecutive chunk positions
      const int per_net = tiles1;
      gemm::gemm_body<BM1, BN1, true, false, gemm::EPI_MUL_DELU>(p1, gemm::xcd_tile_index(raw % per_net, per_net), raw / per_net, smem);
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0, r.t1 = wall_clock64();
    r.hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID
    r.xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) | (kind << 8);    // HW_REG_XCC_ID
    rec[b] = r;
  }
}

template <int BM, int BN>
int tiles_of(const Params& p) { return ((p.J + BN - 1) / BN) * ((p.I + BM - 1) / BM); }

template <int BM0, int BN0, int BM1, int BN1, int OCC = 1>
void variant(const char* name, Params pw, Params px, int splits, int order, int stride, size_t lds_pad, Rec* drec,
             int tpb = 1) {
  const int M = pw.Kc;
  int per = (M + splits - 1) / splits;
  per = (per + 15) / 16 * 16;
  pw.splits = (M + per - 1) / per;
  pw.kc_per_split = per;
  const int t0 = tiles_of<BM0, BN0>(pw), n0 = t0 * pw.nets * pw.splits;
  const int t1 = tiles_of<BM1, BN1>(px), n1 = t1 * px.nets / tpb;
  const size_t l0 = gemm::smem_bytes<BM0, BN0, false, false>(), l1 = gemm::smem_bytes<BM1, BN1, true, false>();
  const size_t lds = std::max(l0, l1) + lds_pad;
  auto kern = pair_tl<BM0, BN0, BM1, BN1, OCC>;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = n0 + n1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kern<<<grid, 256, lds>>>(pw, px, t0, n0, t1, n1, order, stride, drec, tpb);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<grid, 256, lds>>>(pw, px, t0, n0, t1, n1, order, stride, drec, tpb);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  std::vector<Rec> rec(grid);
  CK(hipMemcpy(rec.data(), drec, sizeof(Rec) * grid, hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& r : rec) tmin = std::min(tmin, r.t0), tmax = std::max(tmax, r.t1);
  // per-CU census
  std::map<unsigned, std::pair<int, int>> cu;   // key -> (#problem0, #problem1)
  std::vector<double> end0, end1, start_all;
  for (auto& r : rec) {
    const unsigned key = ((r.xcc & 0xf) << 16) | ((r.hw >> 8) & 0xff);   // xcc | se/sh/cu
    const int kind = (r.xcc >> 8) & 1;
    (kind ? cu[key].second : cu[key].first)++;
    (kind ? end1 : end0).push_back((r.t1 - tmin) * 0.01);
    start_all.push_back((r.t0 - tmin) * 0.01);
  }
  std::map<std::pair<int, int>, int> hist;
  for (auto& kv : cu) hist[kv.second]++;
  auto pct = [](std::vector<double>& v, double q) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[(size_t)(q * (v.size() - 1))];
  };
  printf("%-34s grid %5d (n0 %4d x %3d slabs, n1 %4d) lds %6zu  %.1f us/launch  span %.1f us\n", name, grid, n0,
         per / 16, n1, lds, ms * 1e3f / reps, (tmax - tmin) * 0.01);
  printf("   CUs seen %zu; (n_dW, n_dX) per CU -> #CUs:", cu.size());
  for (auto& kv : hist) printf(" (%d,%d):%d", kv.first.first, kv.first.second, kv.second);
  printf("\n   start p50/p99/max %.1f/%.1f/%.1f us | dW end p10/p50/p90/max %.1f/%.1f/%.1f/%.1f | dX end p10/p50/p90/max %.1f/%.1f/%.1f/%.1f\n",
         pct(start_all, 0.5), pct(start_all, 0.99), pct(start_all, 1.0), pct(end0, 0.1), pct(end0, 0.5), pct(end0, 0.9),
         pct(end0, 1.0), pct(end1, 0.1), pct(end1, 0.5), pct(end1, 0.9), pct(end1, 1.0));
}

void fwd(const char* name, const Params& p) {
  auto kern = gemm::gemm_f32_kernel<128, 128, true, true, gemm::EPI_BIAS_ELU>;
  dim3 grid(((p.J + 127) / 128) * ((p.I + 127) / 128), 1, p.nets);
  const int nwg = grid.x * grid.z;
  const size_t lds = gemm::smem_bytes<128, 128, true, true>();
  unsigned long long* tl;
  CK(hipMalloc(&tl, (size_t)nwg * 4 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(gemm::g_tl), &tl, sizeof(tl)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kern<<<grid, 256, lds>>>(p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) kern<<<grid, 256, lds>>>(p);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)nwg * 4);
  CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
  double loop0 = 0, loop1 = 0;
  for (unsigned b = 0; b < grid.x; ++b) {
    loop0 += h[4 * b + 2] - h[4 * b + 1];
    loop1 += h[4 * (b + grid.x) + 2] - h[4 * (b + grid.x) + 1];
  }
  const double sc = 1.0 / grid.x / 2390.0;
  printf("%-34s %.1f us/launch | main loop: net 0 workgroups %.1f us, net 1 workgroups %.1f us\n", name,
         ms * 1e3f / 20, loop0 * sc, loop1 * sc);
  CK(hipFree(tl));
}

// Empirical fp32-MFMA ceiling: every wave issues independent v_mfma_f32_32x32x2_f32 back to back, no memory traffic.
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters, unsigned long long* clk) {
  gemm::f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  float sacc = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) sacc += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
  if (threadIdx.x == 0 && blockIdx.x < 1024) clk[2 * blockIdx.x] = w1 - w0, clk[2 * blockIdx.x + 1] = c1 - c0;
}

void peak(int blocks_per_cu, int iters) {
  float* out;
  unsigned long long* clk;
  const int grid = 256 * blocks_per_cu;
  CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  CK(hipMalloc(&clk, 2048 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  mfma_peak<<<grid, 256>>>(out, iters, clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) mfma_peak<<<grid, 256>>>(out, iters, clk);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[4];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double flop = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2 * 1);   // waves x MFMAs x flop
  printf("mfma_peak: %d wg/CU x %d iters: %.1f us/launch, %.1f TFLOP/s; block 0: %.1f us wall, shader clock %.0f MHz\n",
         blocks_per_cu, iters, ms * 1e3 / 5, flop / (ms * 1e-3 / 5) * 1e-12, h[0] * 0.01, (double)h[1] / (h[0] * 0.01));
  CK(hipFree(out));
  CK(hipFree(clk));
}

// Round 4: shader clock and MFMA pace INSIDE the slab loops of the production paired launch (dW 128x128 x 32 splits first,
// dX 64x128), from gemm_body's own stamps (wall clock + s_memtime at loop start / end of every workgroup).
void pair_clock(Params pw, Params px) {
  const int M = pw.Kc, splits = 32;
  int per = (M + splits - 1) / splits;
  per = (per + 15) / 16 * 16;
  pw.splits = (M + per - 1) / per, pw.kc_per_split = per;
  const int t0 = tiles_of<128, 128>(pw), n0 = t0 * pw.nets * pw.splits;
  const int t1 = tiles_of<64, 128>(px), n1 = t1 * px.nets;
  const size_t lds = std::max(gemm::smem_bytes<128, 128, false, false>(), gemm::smem_bytes<64, 128, true, false>());
  auto kern = gemm::gemm_pair_kernel<128, 128, false, false, gemm::EPI_PARTIAL, 64, 128, true, false, gemm::EPI_MUL_DELU>;
  const int grid = n0 + n1;
  unsigned long long* tl;
  CK(hipMalloc(&tl, (size_t)grid * 4 * 8));
  CK(hipMemset(tl, 0, (size_t)grid * 4 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(gemm::g_tl), &tl, sizeof(tl)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) kern<<<grid, 256, lds>>>(pw, px, t0, n0, t1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) kern<<<grid, 256, lds>>>(pw, px, t0, n0, t1);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)grid * 4);
  CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  for (int kind = 0; kind < 2; ++kind) {
    std::vector<double> us, ghz, tpm;
    const int lo = kind == 0 ? 0 : n0, hi = kind == 0 ? n0 : grid;
    const double mfmas = kind == 0 ? (per / 16) * 32.0 : (px.Kc / 16) * 16.0;      // per wave: slabs x MFMAs per slab
    for (int b = lo; b < hi; ++b) {
      const double w = (h[4 * b + 3] - h[4 * b + 0]) * 0.01, t = (double)(h[4 * b + 2] - h[4 * b + 1]);
      if (w <= 0) continue;
      us.push_back(w), ghz.push_back(t / w / 1e3), tpm.push_back(t / mfmas);
    }
    printf("pair_clock %s workgroups: main loop %.1f us (median), shader clock %.3f GHz, %.0f MFMAs per wave = %.1f ticks each\n",
           kind == 0 ? "dW (128x128, 32 slabs)" : "dX (64x128, 16 slabs)", med(us), med(ghz), mfmas, med(tpm));
  }
  printf("pair_clock: %.1f us per launch (20 back to back), %.1f TFLOP/s\n", ms * 1e3f / 20,
         (2.0 * 2 * 2 * (double)M * 256 * 256) / (ms * 1e-3 / 20) * 1e-12);
  CK(hipFree(tl));
}

int main() {
  peak(2, 2000);
  const int M = 16384, H = 256;
  const size_t big = (size_t)M * H;
  float *dZ[2], *Hin[2], *W[2], *dX[2], *part, *dbp;
  std::vector<float> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  for (int n = 0; n < 2; ++n) {
    CK(hipMalloc(&dZ[n], big * 4));
    CK(hipMalloc(&Hin[n], big * 4));
    CK(hipMalloc(&dX[n], big * 4));
    CK(hipMalloc(&W[n], H * H * 4));
    CK(hipMemcpy(dZ[n], h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Hin[n], h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W[n], h.data(), H * H * 4, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&part, (size_t)2 * 128 * H * H * 4));
  CK(hipMalloc(&dbp, (size_t)2 * 128 * H * 4));
  Rec* drec;
  CK(hipMalloc(&drec, sizeof(Rec) * 8192));
  Params pw{}, px{};
  pw.nets = 2, pw.I = H, pw.J = H, pw.Kc = M, pw.lda = H, pw.ldb = H, pw.ldc = H;
  pw.c_split_stride = 2 * (int64_t)H * H;
  px.nets = 2, px.splits = 1, px.I = M, px.J = H, px.Kc = H, px.lda = H, px.ldb = H, px.ldc = H, px.ldaux = H;
  for (int n = 0; n < 2; ++n) {
    pw.op[n].A = dZ[n], pw.op[n].B = Hin[n], pw.op[n].C = part + (size_t)n * H * H;
    pw.op[n].dbias = dbp + (size_t)n * 128 * H;
    px.op[n].A = dZ[n], px.op[n].B = W[n], px.op[n].C = dX[n], px.op[n].aux = Hin[n];
  }
  {
    Params pf{};
    float* bias;
    CK(hipMalloc(&bias, 4096));
    CK(hipMemset(bias, 0, 4096));
    pf.nets = 2, pf.splits = 1, pf.I = M, pf.J = H, pf.Kc = H, pf.lda = H, pf.ldb = H, pf.ldc = H;
    for (int n = 0; n < 2; ++n) pf.op[n].A = Hin[n], pf.op[n].B = W[n], pf.op[n].C = dX[n], pf.op[n].bias = bias;
    for (int r = 0; r < 3; ++r) fwd("forward 128x128", pf);
  }
  pair_clock(pw, px);
  return 0;
}

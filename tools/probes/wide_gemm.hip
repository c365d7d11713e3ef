// A/B harness for the wide-wave-tile GEMM (csrc/gemm_bf16_wide.hpp) against the shipped 128x128 kernel (csrc/gemm_bf16.hpp):
// same inputs; the two kernels use different matrix-instruction shapes (32x32x16 vs 16x16x32: the sums inside an instruction
// associate differently), so the results agree to fp32 round-off, not bit for bit: both are checked against fp64.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../non-autoregressive-video-captioning_amd/csrc wide_gemm.hip -o wide_gemm
//   run:   ./wide_gemm M N K [reps]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#define WIDE_TRACE 1
#ifndef WMT
#define WMT 2
#endif
#include "gemm_bf16_wide.hpp"

void nacf_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) / 8388608.0f) - 1.0f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 7680, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 2048;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const int ldc = (N + 3) / 4 * 4;
  std::vector<float> hq((size_t)M * K), hw((size_t)N * K);
  uint64_t s = 12345;
  for (auto& v : hq) v = frand(s);
  for (auto& v : hw) v = frand(s) * 0.05f;
  float *Q, *W, *C0, *C1;
  CK(hipMalloc(&Q, hq.size() * 4)); CK(hipMalloc(&W, hw.size() * 4));
  CK(hipMalloc(&C0, (size_t)M * ldc * 4)); CK(hipMalloc(&C1, (size_t)M * ldc * 4));
  CK(hipMemcpy(Q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const int tiles_k = (K + 31) / 32;
  const int64_t plane = (int64_t)tiles_k * N * 32;
  unsigned short* img; CK(hipMalloc(&img, plane * 3 * 2));
  WImageDesc d{}; d.w = W; d.img = img; d.imgT = nullptr; d.ld = K; d.plane = plane; d.planeT = 0; d.N = N; d.K = K; d.tile0 = 0; d.tiles_k = tiles_k;
  WImageDesc* dd; CK(hipMalloc(&dd, sizeof(d))); CK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(wimage_refresh_kernel<3>, dim3(((N + 31) / 32) * tiles_k), dim3(256), 0, 0, dd, 1);
  CK(hipDeviceSynchronize());

  GemmShape g{};
  g.Q = Q; g.P = W; g.ldq = K; g.ldp = K; g.M = M; g.N = N; g.K = K; g.k_per_split = K;
  g.Pimg = img; g.ldpi = (int64_t)N * 32; g.pimg_plane = plane;
  EpiStore e0{C0, ldc, 0.f, 0, 1}, e1{C1, ldc, 0.f, 0, 1};

  auto kold = gemm_bf16_kernel<128, 128, SRC_F32_KC, SRC_BF16_KC, 3, 3, EpiStore>;
  constexpr size_t old_lds = (size_t)gemm_bf16_lds_chunks<128, 128, 3, 3, false>() * 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kold), hipFuncAttributeMaxDynamicSharedMemorySize, (int)old_lds));
  auto knew = wide::gemm_wide_kernel<WMT, EpiStore>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(knew), hipFuncAttributeMaxDynamicSharedMemorySize, wide::Geo<WMT>::LDS_BYTES));

  GemmShape g0 = g, g1 = g;
  g0.tiles_m = (M + 127) / 128; g0.tiles_n = (N + 127) / 128; if (g0.tiles_n >= 32) g0.group_n = 6;
  g1.tiles_m = (M + wide::Geo<WMT>::BM - 1) / wide::Geo<WMT>::BM; g1.tiles_n = (N + wide::BN - 1) / wide::BN; if (g1.tiles_n >= 16) g1.group_n = 3;
  auto run_old = [&]() { hipLaunchKernelGGL(kold, dim3(g0.tiles_m * g0.tiles_n), dim3(256), old_lds, 0, g0, e0); };
  auto run_new = [&]() { hipLaunchKernelGGL(knew, dim3(g1.tiles_m * g1.tiles_n), dim3(256), wide::Geo<WMT>::LDS_BYTES, 0, g1, e1); };

  CK(hipMemset(C0, 0xff, (size_t)M * ldc * 4)); CK(hipMemset(C1, 0xee, (size_t)M * ldc * 4));
  run_old(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  run_new(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  std::vector<float> h0((size_t)M * ldc), h1((size_t)M * ldc);
  CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0; double maxd = 0; size_t first = (size_t)-1;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      const size_t i = (size_t)m * ldc + n;
      if (memcmp(&h0[i], &h1[i], 4) != 0) { if (first == (size_t)-1) first = i; ++bad; maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i])); }
    }
  {
    size_t nbad = 0; int shown = 0; size_t hist_r[128] = {0}, hist_c[256] = {0};
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { const size_t i = (size_t)m * ldc + n; if (fabsf(h0[i] - h1[i]) > 1e-4f) { ++nbad; hist_r[m % 128]++; hist_c[n % 256]++; if (shown++ < 12) printf("   bad (%d, %d): old %g new %g\n", m, n, h0[i], h1[i]); } }
    printf("   entries off by > 1e-4: %zu; by row %% 128:", nbad); for (int i = 0; i < 128; ++i) if (hist_r[i]) printf(" %d:%zu", i, hist_r[i]);
    printf("\n   by col %% 256:"); for (int i = 0; i < 256; ++i) if (hist_c[i]) printf(" %d:%zu", i, hist_c[i]); printf("\n");
  }
  // fp64 check of both kernels on 4096 sampled entries (error relative to sum |q||w| of the entry)
  double maxe = 0, maxe_old = 0;
  for (int t = 0; t < 4096; ++t) {
    const int m = (int)(((uint64_t)t * 7919 + t / 7) % M), n = (int)(((uint64_t)t * 104729 + t / 3) % N);
    double r = 0, sa = 0; for (int k = 0; k < K; ++k) { const double pr = (double)hq[(size_t)m * K + k] * hw[(size_t)n * K + k]; r += pr; sa += fabs(pr); }
    maxe = std::max(maxe, fabs(r - h1[(size_t)m * ldc + n]) / sa);
    maxe_old = std::max(maxe_old, fabs(r - h0[(size_t)m * ldc + n]) / sa);
  }
  printf("M=%d N=%d K=%d: new vs old: %zu of %zu entries differ in some bit, max |diff| %.3g; error vs fp64 / sum|q w|: new %.3g, old %.3g  %s\n", M, N, K, bad,
         (size_t)M * N, maxd, maxe, maxe_old, (maxe < 2e-7 && maxd < 1e-4) ? "OK" : "*** WRONG ***");

  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> t0, t1;
  for (int r = 0; r < reps; ++r) {
    float ms;
    CK(hipEventRecord(a)); run_old(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); t0.push_back(ms);
    CK(hipEventRecord(a)); run_new(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); t1.push_back(ms);
  }
  std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end());
  const double fl = 2.0 * M * N * K;
  printf("  old 128x128: median %.1f us (min %.1f) = %.1f TF   |   wide 128x256: median %.1f us (min %.1f) = %.1f TF   (wgs %d vs %d)\n",
         t0[reps / 2] * 1e3, t0[0] * 1e3, fl / (t0[reps / 2] * 1e-3) * 1e-12, t1[reps / 2] * 1e3, t1[0] * 1e3, fl / (t1[reps / 2] * 1e-3) * 1e-12,
         g0.tiles_m * g0.tiles_n, g1.tiles_m * g1.tiles_n);
  // ablations + phase stamps (wave 0 of every workgroup)
  unsigned long long* tr; CK(hipMalloc(&tr, 4096 * 4 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(wide::g_wide_trace), &tr, sizeof(tr)));
  const int nwg1 = g1.tiles_m * g1.tiles_n;
  auto variant = [&](const char* name, auto kern) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, wide::Geo<WMT>::LDS_BYTES));
    std::vector<float> tt;
    for (int r = 0; r < reps; ++r) {
      float ms;
      CK(hipEventRecord(a)); hipLaunchKernelGGL(kern, dim3(nwg1), dim3(256), wide::Geo<WMT>::LDS_BYTES, 0, g1, e1); CK(hipEventRecord(b));
      CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); tt.push_back(ms);
    }
    std::sort(tt.begin(), tt.end());
    std::vector<unsigned long long> h(4 * (size_t)std::min(nwg1, 4096));
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0; const int n = (int)h.size() / 4;
    for (int i = 0; i < n; ++i) { pro += h[4 * i + 1] - h[4 * i]; loop += h[4 * i + 2] - h[4 * i + 1]; epi += h[4 * i + 3] - h[4 * i + 2]; }
    printf("  %-28s median %.1f us = %.1f TF | per workgroup: prologue %.0f, loop %.0f (%.0f per k-tile; %d = MFMA), epilogue %.0f cycles\n", name,
           tt[reps / 2] * 1e3, fl / (tt[reps / 2] * 1e-3) * 1e-12, pro / n, loop / n, loop / n / (K / 32), 1536 * WMT, epi / n);
  };
  variant("wide", wide::gemm_wide_kernel<WMT, EpiStore, 0>);
  for (int skew : {8, 16, 32, 64}) {
    CK(hipMemcpyToSymbol(HIP_SYMBOL(wide::g_wide_skew), &skew, sizeof(skew)));
    char nm[64]; snprintf(nm, sizeof(nm), "wide, start skew %d x 64 cycles", skew);
    variant(nm, wide::gemm_wide_kernel<WMT, EpiStore, 0>);
  }
  { int z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(wide::g_wide_skew), &z, sizeof(z))); }
  variant("wide, no DMA in loop", wide::gemm_wide_kernel<WMT, EpiStore, 1>);
  variant("wide, no split", wide::gemm_wide_kernel<WMT, EpiStore, 2>);
  variant("wide, no DMA, no split", wide::gemm_wide_kernel<WMT, EpiStore, 3>);
  return (maxe < 2e-7 && maxd < 1e-4) ? 0 : 2;
}

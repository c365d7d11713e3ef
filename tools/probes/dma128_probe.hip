// A/B + ablation harness for the DMA-fed two-per-CU exact-mode GEMM (csrc/gemm_dma128.hpp) against the register-staged 128x128
// kernel (csrc/gemm_bf16.hpp): timing, per-workgroup phase stamps (prologue / k-loop / epilogue) and ablations (no DMA in the
// loop, no operand split, no matrix instructions).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDMT=2 -I../../non-autoregressive-video-captioning_amd/csrc -I../../include dma128_probe.hip -o dma128_probe_2
//   run:   ./dma128_probe_2 M N K [reps]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#define DMA128_TRACE 1
#ifndef DMT
#define DMT 2
#endif
#include "gemm_dma128.hpp"

void nacf_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) / 8388608.0f) - 1.0f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 5120, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
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
  g.Q = Q; g.P = W; g.ldq = K; g.ldp = K; g.M = M; g.N = N; g.K = K; g.k_per_split = (K + 31) / 32 * 32;
  g.Pimg = img; g.ldpi = (int64_t)N * 32; g.pimg_plane = plane;
  EpiStore e0{C0, ldc, 0.f, 0, 1}, e1{C1, ldc, 0.f, 0, 1};

  auto kold = gemm_bf16_kernel<128, 128, SRC_F32_KC, SRC_BF16_KC, 3, 3, EpiStore>;
  constexpr size_t old_lds = (size_t)gemm_bf16_lds_chunks<128, 128, 3, 3, false>() * 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kold), hipFuncAttributeMaxDynamicSharedMemorySize, (int)old_lds));
  using G = dma128::Geo<DMT>;
  GemmShape g0 = g, g1 = g;
  g0.tiles_m = (M + 127) / 128; g0.tiles_n = (N + 127) / 128; if (g0.tiles_n >= 32) g0.group_n = 6;
  g1.tiles_m = (M + G::BM - 1) / G::BM; g1.tiles_n = (N + dma128::BN - 1) / dma128::BN; if (g1.tiles_n >= 32) g1.group_n = 6;
  const int nwg1 = g1.tiles_m * g1.tiles_n;
  auto run_old = [&]() { hipLaunchKernelGGL(kold, dim3(g0.tiles_m * g0.tiles_n), dim3(256), old_lds, 0, g0, e0); };
  auto knew = dma128::gemm_dma128_kernel<DMT, EpiStore, 0>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(knew), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  auto run_new = [&]() { hipLaunchKernelGGL(knew, dim3(nwg1), dim3(256), G::LDS_BYTES, 0, g1, e1); };
  {
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(knew), 256, G::LDS_BYTES));
    printf("occupancy: %d workgroups per CU (LDS %d bytes)\n", nb, G::LDS_BYTES);
  }

  CK(hipMemset(C0, 0xff, (size_t)M * ldc * 4)); CK(hipMemset(C1, 0xee, (size_t)M * ldc * 4));
  run_old(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  run_new(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  std::vector<float> h0((size_t)M * ldc), h1((size_t)M * ldc);
  CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
  double maxd = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) { const size_t i = (size_t)m * ldc + n; maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i])); }
  double maxe = 0;
  for (int t = 0; t < 2048; ++t) {
    const int m = (int)(((uint64_t)t * 7919 + t / 7) % M), n = (int)(((uint64_t)t * 104729 + t / 3) % N);
    double r = 0, sa = 0; for (int k = 0; k < K; ++k) { const double pr = (double)hq[(size_t)m * K + k] * hw[(size_t)n * K + k]; r += pr; sa += fabs(pr); }
    maxe = std::max(maxe, fabs(r - h1[(size_t)m * ldc + n]) / sa);
  }
  printf("M=%d N=%d K=%d MT=%d: max |new - old| %.3g; error vs fp64 / sum|q w| %.3g  %s\n", M, N, K, DMT, maxd, maxe, (maxe < 2e-7 && maxd < 1e-4) ? "OK" : "*** WRONG ***");

  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> t0, t1;
  for (int r = 0; r < reps; ++r) {
    float ms;
    CK(hipEventRecord(a)); run_old(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); t0.push_back(ms);
    CK(hipEventRecord(a)); run_new(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); t1.push_back(ms);
  }
  std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end());
  const double fl = 2.0 * M * N * K;
  printf("  old 128x128: median %.1f us (min %.1f) = %.1f TF   |   dma128: median %.1f us (min %.1f) = %.1f TF   (wgs %d vs %d)\n",
         t0[reps / 2] * 1e3, t0[0] * 1e3, fl / (t0[reps / 2] * 1e-3) * 1e-12, t1[reps / 2] * 1e3, t1[0] * 1e3, fl / (t1[reps / 2] * 1e-3) * 1e-12,
         g0.tiles_m * g0.tiles_n, nwg1);
  unsigned long long* tr; CK(hipMalloc(&tr, 8192 * 4 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(dma128::g_trace), &tr, sizeof(tr)));
  const int nk = (K + 31) / 32;
  auto variant = [&](const char* name, auto kern) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    std::vector<float> tt;
    for (int r = 0; r < reps; ++r) {
      float ms;
      CK(hipEventRecord(a)); hipLaunchKernelGGL(kern, dim3(std::min(nwg1, 8192)), dim3(256), G::LDS_BYTES, 0, g1, e1); CK(hipEventRecord(b));
      CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); tt.push_back(ms);
    }
    std::sort(tt.begin(), tt.end());
    std::vector<unsigned long long> h(4 * (size_t)std::min(nwg1, 8192));
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0; const int n = (int)h.size() / 4;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < n; ++i) {
      pro += h[4 * i + 1] - h[4 * i]; loop += h[4 * i + 2] - h[4 * i + 1]; epi += h[4 * i + 3] - h[4 * i + 2];
      tmin = std::min(tmin, h[4 * i]); tmax = std::max(tmax, h[4 * i + 3]);
    }
    printf("  %-30s median %.1f us = %.1f TF | per workgroup: prologue %.0f, loop %.0f (%.0f per k-tile; %d = MFMA alone), epilogue %.0f; launch span %.0f k cycles\n",
           name, tt[reps / 2] * 1e3, fl / (tt[reps / 2] * 1e-3) * 1e-12, pro / n, loop / n, loop / n / nk, 768 * DMT, epi / n, (double)(tmax - tmin) / 1000.0);
  };
  variant("dma128", dma128::gemm_dma128_kernel<DMT, EpiStore, 0>);
  variant("dma128, no DMA in loop", dma128::gemm_dma128_kernel<DMT, EpiStore, 1>);
  variant("dma128, no split", dma128::gemm_dma128_kernel<DMT, EpiStore, 2>);
  variant("dma128, no DMA, no split", dma128::gemm_dma128_kernel<DMT, EpiStore, 3>);
  variant("dma128, no MFMA", dma128::gemm_dma128_kernel<DMT, EpiStore, 4>);
  variant("dma128, no MFMA, no split", dma128::gemm_dma128_kernel<DMT, EpiStore, 6>);
  // ---- the FFN1 epilogue (bias, gelu_new, pre-activation store, dropout, residual-free) and the projection epilogue (bias, dropout,
  //      residual): old kernel vs this one, with the phase stamps
  {
    float *bias, *pre, *res; uint64_t* rng;
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&pre, (size_t)M * ldc * 4)); CK(hipMalloc(&res, (size_t)M * ldc * 4)); CK(hipMalloc(&rng, 16));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(res, 0, (size_t)M * ldc * 4));
    uint64_t hr[2] = {1234, 7}; CK(hipMemcpy(rng, hr, 16, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 2; ++kind) {
      EpiLinear el; memset(&el, 0, sizeof(el));
      el.Y = C1; el.ldy = ldc; el.vec_out = 1; el.vec_bias = 1;
      el.ep.bias = bias; el.ep.rng_state = rng;
      if (kind == 0) { el.ep.act = NACF_ACT_GELU_NEW; el.ep.preact = pre; el.ep.ld_preact = ldc; el.ep.p_drop1 = 0.5f; el.ep.salt1 = 3; }
      else { el.ep.act = NACF_ACT_NONE; el.ep.p_drop1 = 0.5f; el.ep.salt1 = 3; el.ep.residual = res; el.ep.ld_residual = ldc; }
      auto ko = gemm_bf16_kernel<128, 128, SRC_F32_KC, SRC_BF16_KC, 3, 3, EpiLinear>;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ko), hipFuncAttributeMaxDynamicSharedMemorySize, (int)old_lds));
      auto ko64 = gemm_bf16_kernel<64, 64, SRC_F32_KC, SRC_BF16_KC, 3, 2, EpiLinear>;
      constexpr size_t old_lds64 = (size_t)gemm_bf16_lds_chunks<64, 64, 3, 2, false>() * 16;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ko64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)old_lds64));
      GemmShape g64 = g; g64.tiles_m = (M + 63) / 64; g64.tiles_n = (N + 63) / 64;
      auto kn = dma128::gemm_dma128_kernel<DMT, EpiLinear, 0>;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
      std::vector<float> ta, tb, tc;
      for (int r = 0; r < reps; ++r) {
        float ms;
        CK(hipEventRecord(a)); hipLaunchKernelGGL(ko, dim3(g0.tiles_m * g0.tiles_n), dim3(256), old_lds, 0, g0, el); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); ta.push_back(ms);
        CK(hipEventRecord(a)); hipLaunchKernelGGL(ko64, dim3(g64.tiles_m * g64.tiles_n), dim3(256), old_lds64, 0, g64, el); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); tc.push_back(ms);
        CK(hipEventRecord(a)); hipLaunchKernelGGL(kn, dim3(nwg1), dim3(256), G::LDS_BYTES, 0, g1, el); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); tb.push_back(ms);
      }
      std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end()); std::sort(tc.begin(), tc.end());
      std::vector<unsigned long long> h(4 * (size_t)std::min(nwg1, 8192));
      CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
      double pro = 0, loop = 0, epi = 0; const int n = (int)h.size() / 4;
      for (int i = 0; i < n; ++i) { pro += h[4 * i + 1] - h[4 * i]; loop += h[4 * i + 2] - h[4 * i + 1]; epi += h[4 * i + 3] - h[4 * i + 2]; }
      printf("  EpiLinear %s: old 128x128 %.1f us, old 64x64 %.1f us, dma128 %.1f us | dma128 per workgroup: prologue %.0f, loop %.0f, epilogue %.0f\n",
             kind == 0 ? "gelu + preact + dropout" : "dropout + residual", ta[reps / 2] * 1e3, tc[reps / 2] * 1e3, tb[reps / 2] * 1e3, pro / n, loop / n, epi / n);
    }
  }
  return 0;
}

// A/B harness for the panel GEMM (csrc/gemm_bf16_panel.hpp) against the shipped 64x64 / 128x128 exact-mode kernels
// (csrc/gemm_bf16.hpp) on the decoder layer's skinny shapes: same inputs, both checked against fp64 (different matrix
// instruction shapes and a different summation tree: agreement to fp32 round-off, not bit for bit).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../non-autoregressive-video-captioning_amd/csrc panel_gemm.hip -o panel_gemm
//   run:   ./panel_gemm M N K [reps] [live_percent]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#ifndef PANEL_TRACE
#define PANEL_TRACE 1
#endif
#ifndef PMT
#define PMT 2
#endif
#ifndef PNT
#define PNT 4
#endif
#include "gemm_bf16_panel.hpp"

void nacf_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) / 8388608.0f) - 1.0f; }
static inline uint32_t fbits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float bitsf(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 5120, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const int live_pct = argc > 5 ? atoi(argv[5]) : 100;
  const int ldc = (N + 3) / 4 * 4;
  if (K % 256 || K < 512 || N % (32 * PNT)) { printf("K must be a multiple of 256 (>= 512), N of %d\n", 32 * PNT); return 1; }
  std::vector<float> hq((size_t)M * K), hw((size_t)N * K);
  uint64_t s = 12345;
  for (auto& v : hq) v = frand(s);
  for (auto& v : hw) v = frand(s) * 0.05f;
  float *Q, *W, *C0, *C1;
  CK(hipMalloc(&Q, hq.size() * 4)); CK(hipMalloc(&W, hw.size() * 4));
  CK(hipMalloc(&C0, (size_t)M * ldc * 4)); CK(hipMalloc(&C1, (size_t)M * ldc * 4));
  CK(hipMemcpy(Q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  // k-tile-major image for the old kernel
  const int tiles_k = (K + 31) / 32;
  const int64_t plane = (int64_t)tiles_k * N * 32;
  unsigned short* img; CK(hipMalloc(&img, plane * 3 * 2));
  WImageDesc d{}; d.w = W; d.img = img; d.imgT = nullptr; d.ld = K; d.plane = plane; d.planeT = 0; d.N = N; d.K = K; d.tile0 = 0; d.tiles_k = tiles_k;
  WImageDesc* dd; CK(hipMalloc(&dd, sizeof(d))); CK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(wimage_refresh_kernel<3>, dim3(((N + 31) / 32) * tiles_k), dim3(256), 0, 0, dd, 1);
  CK(hipDeviceSynchronize());
  // fragment-major image, built on the host (the library builds it in wimage_refresh_kernel; this is the independent statement)
  const int NT32 = (N + 31) / 32, K16 = K / 16;
  const int64_t ldpf = (int64_t)NT32 * 3 * 512;
  std::vector<unsigned short> hf((size_t)K16 * ldpf, 0);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      float x = hw[(size_t)n * K + k];
      unsigned short t[3];
      for (int p = 0; p < 3; ++p) { const uint32_t top = fbits(x) & 0xffff0000u; t[p] = (unsigned short)(top >> 16); x -= bitsf(top); }
      const int k16 = k / 16, nt = n / 32, h = (k & 15) >> 3, e = k & 7;
      for (int p = 0; p < 3; ++p) hf[(size_t)k16 * ldpf + ((size_t)(nt * 3 + p) * 64 + h * 32 + (n & 31)) * 8 + e] = t[p];
    }
  unsigned short* fimg; CK(hipMalloc(&fimg, hf.size() * 2)); CK(hipMemcpy(fimg, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  // live-row list: a random subset in ascending order, then the dead ones
  std::vector<int> rows(M); int n_live = 0;
  { std::vector<int> dead; for (int m = 0; m < M; ++m) { if ((int)((frand(s) + 1.f) * 50.f) < live_pct) rows[n_live++] = m; else dead.push_back(m); }
    for (size_t i = 0; i < dead.size(); ++i) rows[n_live + i] = dead[i]; }
  int *drows = nullptr, *dcount = nullptr;
  if (live_pct < 100) { CK(hipMalloc(&drows, M * 4)); CK(hipMalloc(&dcount, 4)); CK(hipMemcpy(drows, rows.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcount, &n_live, 4, hipMemcpyHostToDevice)); }
  else n_live = M;

  GemmShape g{};
  g.Q = Q; g.P = W; g.ldq = K; g.ldp = K; g.M = M; g.N = N; g.K = K; g.k_per_split = K;
  g.Pimg = img; g.ldpi = (int64_t)N * 32; g.pimg_plane = plane;
  g.Pfrag = fimg; g.ldpf = ldpf;
  g.rows = drows; g.count = dcount; g.zero_dead = drows ? 1 : 0;
  EpiStore e0{C0, ldc, 0.f, 0, 1}, e1{C1, ldc, 0.f, 0, 1};

  using PG = panel::Geo<PMT, PNT, 3>;
  auto kold64 = gemm_bf16_kernel<64, 64, SRC_F32_KC, SRC_BF16_KC, 3, 2, EpiStore>;
  constexpr size_t old_lds = (size_t)gemm_bf16_lds_chunks<64, 64, 3, 2, false>() * 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kold64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)old_lds));
  GemmShape g0 = g, g1 = g;
  g0.tiles_m = (M + 63) / 64; g0.tiles_n = (N + 63) / 64;
  g1.tiles_m = (M + PG::BM - 1) / PG::BM; g1.tiles_n = N / PG::BN;
  const int grid0 = (g0.tiles_m + (g.zero_dead ? 1 : 0)) * g0.tiles_n, grid1 = (g1.tiles_m + (g.zero_dead ? 1 : 0)) * g1.tiles_n;
  auto run_old = [&]() { hipLaunchKernelGGL(kold64, dim3(grid0), dim3(256), old_lds, 0, g0, e0); };
  const int pgrid = getenv("PANEL_GRID") ? atoi(getenv("PANEL_GRID")) : 256;
  auto kp = panel::gemm_panel_kernel<PMT, PNT, 3, EpiStore>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, PG::LDS_BYTES));
  const int ptiles = g1.tiles_m * g1.tiles_n;
  const int pg = ptiles >= pgrid ? pgrid : (ptiles + 7) / 8 * 8;
  auto run_new = [&]() { hipLaunchKernelGGL(kp, dim3(pg), dim3(256), PG::LDS_BYTES, 0, g1, e1); };

  CK(hipMemset(C0, 0xff, (size_t)M * ldc * 4)); CK(hipMemset(C1, 0xee, (size_t)M * ldc * 4));
  run_old(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  run_new(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
  std::vector<float> h0((size_t)M * ldc), h1((size_t)M * ldc);
  CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
  size_t nbad = 0; double maxd = 0; int shown = 0; size_t hist_r[64] = {0}, hist_c[128] = {0};
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      const size_t i = (size_t)m * ldc + n;
      const double df = fabs((double)h0[i] - h1[i]);
      if (!(df <= 1e-4)) { ++nbad; hist_r[m % 64]++; hist_c[n % 128]++; if (shown++ < 10) printf("   bad (%d, %d): old %g new %g\n", m, n, h0[i], h1[i]); }
      else maxd = std::max(maxd, df);
    }
  if (nbad) {
    printf("   entries off by > 1e-4: %zu; by row %% 64:", nbad); for (int i = 0; i < 64; ++i) if (hist_r[i]) printf(" %d:%zu", i, hist_r[i]);
    printf("\n   by col %% 128:"); for (int i = 0; i < 128; ++i) if (hist_c[i]) printf(" %d:%zu", i, hist_c[i]); printf("\n");
  }
  std::vector<char> is_live(M, 0); for (int i = 0; i < n_live; ++i) is_live[rows[i]] = 1;
  double maxe = 0, maxe_old = 0;
  for (int t = 0; t < 4096; ++t) {
    const int m = (int)(((uint64_t)t * 7919 + t / 7) % M), n = (int)(((uint64_t)t * 104729 + t / 3) % N);
    double r = 0, sa = 0;
    if (is_live[m]) for (int k = 0; k < K; ++k) { const double pr = (double)hq[(size_t)m * K + k] * hw[(size_t)n * K + k]; r += pr; sa += fabs(pr); }
    else sa = 1;
    maxe = std::max(maxe, fabs(r - h1[(size_t)m * ldc + n]) / sa);
    maxe_old = std::max(maxe_old, fabs(r - h0[(size_t)m * ldc + n]) / sa);
  }
  const bool ok = maxe < 2e-7 && nbad == 0;
  printf("M=%d (live %d) N=%d K=%d: new vs old: max |diff| %.3g (%zu entries off by > 1e-4); error vs fp64 / sum|q w|: new %.3g, old %.3g  %s\n", M, n_live, N, K,
         maxd, nbad, maxe, maxe_old, ok ? "OK" : "*** WRONG ***");

  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> t0, t1;
  for (int r = 0; r < reps; ++r) {
    float ms;
    CK(hipEventRecord(a)); run_old(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); t0.push_back(ms);
    CK(hipEventRecord(a)); run_new(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); t1.push_back(ms);
  }
  std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end());
  // back to back (no event pair per launch): the time a dependent chain of such launches sees
  float bb0, bb1;
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) run_old(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&bb0, a, b));
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) run_new(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&bb1, a, b));
  const double fl = 2.0 * n_live * N * K;
  printf("  old 64x64: median %.1f us (min %.1f, back to back %.1f) = %.1f TF   |   panel %dx%d: median %.1f us (min %.1f, back to back %.1f) = %.1f TF   (wgs %d vs %d)\n",
         t0[reps / 2] * 1e3, t0[0] * 1e3, bb0 / reps * 1e3, fl / (t0[reps / 2] * 1e-3) * 1e-12, PG::BM, PG::BN, t1[reps / 2] * 1e3, t1[0] * 1e3, bb1 / reps * 1e3,
         fl / (t1[reps / 2] * 1e-3) * 1e-12, grid0, pg);
  // phase stamps (wave 0 of every workgroup)
  unsigned long long* tr; CK(hipMalloc(&tr, 2 * 8192 * 8 * 8)); CK(hipMemset(tr, 0, 2 * 8192 * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(panel::g_panel_trace), &tr, sizeof(tr)));
  run_new(); CK(hipDeviceSynchronize());
  {
    const int nlive_wg = ((n_live + PG::BM - 1) / PG::BM) * g1.tiles_n;
    const int n = std::min(nlive_wg, 8192);
    std::vector<unsigned long long> h(8 * (size_t)n);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    int per_xcc[8] = {0};
    for (int i = 0; i < n; ++i) {
      loop += h[8 * i + 2] - h[8 * i]; epi += h[8 * i + 3] - h[8 * i + 2];
      tmin = std::min(tmin, h[8 * i + 4]); tmax = std::max(tmax, h[8 * i + 5]); per_xcc[h[8 * i + 6] & 7]++;
    }
    const int steps = K / 64;      // k-steps per wave
    printf("  per tile (wave 0, %d tiles): loop %.0f (%.0f per k-step; %d = matrix instructions), epilogue %.0f cycles; first start -> last end %.2f us\n",
           n, loop / n, loop / n / steps, 32 * PG::NSLOT, epi / n, (tmax - tmin) * 0.01);
    std::vector<double> st, en, du;
    for (int i = 0; i < n; ++i) { st.push_back((h[8 * i + 4] - tmin) * 0.01); en.push_back((h[8 * i + 5] - tmin) * 0.01); du.push_back((h[8 * i + 5] - h[8 * i + 4]) * 0.01); }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
    auto pc = [&](std::vector<double>& v, int p) { return v[std::min(n - 1, n * p / 100)]; };
    printf("    start us at 0/10/25/50/75/90/100 %%: %.2f %.2f %.2f %.2f %.2f %.2f %.2f;  end: %.2f %.2f %.2f %.2f %.2f %.2f %.2f;  duration: %.2f %.2f %.2f %.2f %.2f %.2f %.2f;  per XCC:",
           pc(st, 0), pc(st, 10), pc(st, 25), pc(st, 50), pc(st, 75), pc(st, 90), pc(st, 100), pc(en, 0), pc(en, 10), pc(en, 25), pc(en, 50), pc(en, 75), pc(en, 90), pc(en, 100),
           pc(du, 0), pc(du, 10), pc(du, 25), pc(du, 50), pc(du, 75), pc(du, 90), pc(du, 100));
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    std::vector<unsigned long long> sc(8 * 256);
    CK(hipMemcpy(sc.data(), tr + 8192 * 8, sc.size() * 8, hipMemcpyDeviceToHost));
    double st4[4] = {0, 0, 0, 0};
    for (int w = 0; w < 256; ++w) if (sc[8 * w]) { for (int i = 0; i < 4; ++i) st4[i] += sc[8 * w + i]; }
    const double trips = (double)n * (K / 256);
    printf("    cycles per k-step by position in the chunk (all %d tiles): %.0f %.0f %.0f %.0f   [PANEL_ABL=%d]\n", n, st4[0] / trips, st4[1] / trips, st4[2] / trips, st4[3] / trips, PANEL_ABL);
  }
  return ok ? 0 : 2;
}

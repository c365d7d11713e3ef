// Probe of the weight-gradient body on fp32 operands (csrc/gemm_g256w.hpp): C[I][J] = A[R][I]^T B[R][J], NS = 1 | 3, dense rows or a row list.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../non-autoregressive-video-captioning_amd/csrc -I ../../include gemm256w_probe.hip -o gemm256w_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "gemm_g256w.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef PHASES
#define PHASES 4
#endif
struct EpiPlain {
  float* C; int64_t ldc; int rows, cols;
  __device__ __forceinline__ void operator()(int r, int c, g256::f32x4 v) const {
    if (r >= rows || c >= cols) return;
    *reinterpret_cast<g256::f32x4*>(C + (int64_t)r * ldc + c) = v;
  }
};
struct EpiNoSum { __device__ __forceinline__ void operator()(int, float) const {} };

template <int NS>
__global__ __launch_bounds__(512, 2) void gemm_w32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, const int* list, int R,
                                                    int I, int J, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_j = (J + 255) / 256, tiles = tiles_j * ((I + 255) / 256);
  const int z = blockIdx.x / tiles, tile = blockIdx.x % tiles, ti = tile / tiles_j, tj = tile % tiles_j;
  g256w::Walk w;
  w.n_live = R; w.list = g256w::row_list(list, R); w.tile0 = z; w.step = S;
  const int nk_all = (R + 31) / 32;
  w.nk = max(1, (nk_all - z + S - 1) / S);
  const g256w::Operand oa = g256w::operand(A, I, ti * 256, R, lane, wave), ob = g256w::operand(B, J, tj * 256, R, lane, wave);
  EpiPlain epi;
  epi.C = C + (size_t)z * I * J + (size_t)ti * 256 * J + tj * 256; epi.ldc = J; epi.rows = I - ti * 256; epi.cols = J - tj * 256;
  g256w::body<NS, false, PHASES>(smem, oa, ob, w, epi, EpiNoSum());
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, g256w::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w32<3>), hipFuncAttributeMaxDynamicSharedMemorySize, g256w::LDS_BYTES));
  const size_t n = (size_t)8192 * 4096;
  float *A, *B, *C; int* L;
  CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&B, n * 4)); CK(hipMalloc(&C, (size_t)4096 * 4096 * 4 * 2)); CK(hipMalloc(&L, 8192 * 4));
  std::vector<float> h(n);
  uint64_t s = 7;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((s >> 40) / 8388608.0f) - 1.0f; }
  CK(hipMemcpy(A, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<int> hl(8192);
  for (int i = 0; i < 8192; ++i) hl[i] = (i * 37) % 4096 + (i / 4096) * 4096;      // a permutation of each half (the cases with a list use 4096 rows)
  CK(hipMemcpy(L, hl.data(), 8192 * 4, hipMemcpyHostToDevice));
  struct Case { const char* name; int R, I, J, S, use_list; };
  const Case cases[] = {{"cube 4096 dense", 4096, 4096, 4096, 1, 0}, {"cube 4096 row list", 4096, 4096, 4096, 1, 1}, {"8192 x 2048 x 2048, 8 splits", 8192, 2048, 2048, 4, 0},
                        {"enc_lin 7680 x 512 x 2048, 16 splits", 7680, 512, 2048, 16, 0},
                        {"vocabulary dX-like: reduce 10560, 2560 x 512, 12 splits", 10560, 2560, 512, 12, 0},
                        {"vocabulary dX-like: reduce 10560, 2560 x 512, 24 splits", 10560, 2560, 512, 24, 0}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int ns = 1; ns <= 3; ns += 2)
    for (const Case& c : cases) {
      const int tiles = ((c.I + 255) / 256) * ((c.J + 255) / 256);
      auto go = [&] {
        if (ns == 1) hipLaunchKernelGGL(gemm_w32<1>, dim3(tiles * c.S), dim3(512), g256w::LDS_BYTES, 0, A, B, C, c.use_list ? L : nullptr, c.R, c.I, c.J, c.S);
        else hipLaunchKernelGGL(gemm_w32<3>, dim3(tiles * c.S), dim3(512), g256w::LDS_BYTES, 0, A, B, C, c.use_list ? L : nullptr, c.R, c.I, c.J, c.S);
      };
      go(); CK(hipDeviceSynchronize());
      if (c.S == 1) {      // sampled fp64 check (A and B hold the same data)
        double worst = 0;
        for (int t = 0; t < 64; ++t) {
          const int i = (t * 997 + 13) % c.I, j = (t * 4051 + 7) % c.J;
          double ref = 0, mag = 0;
          for (int r = 0; r < c.R; ++r) {
            const int pr = c.use_list ? hl[r] : r;
            const double x = (double)h[(size_t)pr * c.I + i] * (double)h[(size_t)pr * c.J + j];
            ref += x; mag += fabs(x);
          }
          float got; CK(hipMemcpy(&got, C + (size_t)i * c.J + j, 4, hipMemcpyDeviceToHost));
          worst = std::max(worst, fabs(got - ref) / mag);
        }
        printf("   check: worst |err| / sum|terms| over 64 samples = %.2e\n", worst);
      }
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) go();
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps, fl = 2.0 * c.R * c.I * c.J;
      printf("PH=%d NS=%d %-40s %9.1f us  %7.1f TF useful  (%d workgroups, %d k-tiles each: %.2f us per k-tile)\n", PHASES, ns, c.name, us, fl / us / 1e6, tiles * c.S,
             (c.R / 32) / c.S, us / ((double)(c.R / 32) / c.S * std::max(1, (tiles * c.S + 255) / 256)));
    }
  return 0;
}

// Probe of the bf16-RESIDENT 256 x 256 x 64 eight-phase GEMM body (csrc/gemm_g256.hpp).
// Standalone: no library, no torch.  Two forms behind one k-loop:
//   F (forward / dX):  C[M][N] (fp32 | bf16) = A[M][K] (bf16) * B[N][K]^T (bf16)        K % 64 == 0, M / N ragged
//   W (dW):            C[I][J] (fp32, split-k slabs) = A[R][I]^T (bf16) * B[R][J] (bf16)  R ragged, transposing LDS reads
//
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../non-autoregressive-video-captioning_amd/csrc gemm256_8phase.hip -o gemm256_8phase
//   run:   ./gemm256_8phase [reps]      (ragged cases against fp64, then the model's shapes and cubes, uniform random operands)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include "gemm_g256.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef GROUP_M
#define GROUP_M 8            // tile rows per L2 group: 32 consecutive logical tiles = 8 x 4
#endif
using g256::f32x4;
constexpr int BM = g256::BM, BN = g256::BN, LDS_BYTES = g256::LDS_BYTES;

template <bool OUT_BF16>
struct EpiPlain {
  void* C; int64_t ldc; int rows, cols;      // rows / cols left from the tile origin
  __device__ __forceinline__ void operator()(int r, int c, f32x4 v) const {
    if (r >= rows) return;
    if constexpr (OUT_BF16) {
      __bf16* cp = reinterpret_cast<__bf16*>(C) + (size_t)r * ldc + c;
      const g256::bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      if (c + 3 < cols && (ldc & 3) == 0) *reinterpret_cast<g256::bf16x4*>(cp) = o;
      else
        for (int e = 0; e < 4; ++e)
          if (c + e < cols) cp[e] = o[e];
    } else {
      float* cp = reinterpret_cast<float*>(C) + (size_t)r * ldc + c;
      if (c + 3 < cols && (ldc & 3) == 0) *reinterpret_cast<f32x4*>(cp) = v;
      else
        for (int e = 0; e < 4; ++e)
          if (c + e < cols) cp[e] = v[e];
    }
  }
};

template <bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_f(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int tm, tn;
  g256::tile_of_block(blockIdx.x, tiles_m * tiles_n, tiles_m, tiles_n, GROUP_M, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const g256::Stage sa = g256::stage_f(A, K, m0, M, 0, lane, wave), sb = g256::stage_f(B, K, n0, N, 0, lane, wave);
  EpiPlain<OUT_BF16> epi;
  epi.C = OUT_BF16 ? (void*)(reinterpret_cast<uint16_t*>(Cv) + (size_t)m0 * N + n0) : (void*)(reinterpret_cast<float*>(Cv) + (size_t)m0 * N + n0);
  epi.ldc = N; epi.rows = M - m0; epi.cols = N - n0;
  g256::body<false>(smem, sa, sb, K / 64, epi);
}

// W form: blockIdx.x = tile, blockIdx.y = split; slab z at C + z * I * J
__global__ __launch_bounds__(512, 2) void gemm_w(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C, int R, int I, int J,
                                                  int kt_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_i = (I + BM - 1) / BM, tiles_j = (J + BN - 1) / BN;
  int ti, tj;
  g256::tile_of_block(blockIdx.x, tiles_i * tiles_j, tiles_i, tiles_j, GROUP_M, ti, tj);
  const int i0 = ti * BM, j0 = tj * BN, z = blockIdx.y;
  const int nk_all = (R + 63) / 64, kt0 = z * kt_per_split, nk = min(kt_per_split, nk_all - kt0);
  const g256::Stage sa = g256::stage_w(A, I, i0, R, kt0, lane, wave), sb = g256::stage_w(B, J, j0, R, kt0, lane, wave);
  EpiPlain<false> epi;
  epi.C = C + (size_t)z * I * J + (size_t)i0 * J + j0; epi.ldc = J; epi.rows = I - i0; epi.cols = J - j0;
  g256::body<true>(smem, sa, sb, nk, epi);
}

static uint16_t f2bf(float x) {      // round to nearest even
  uint32_t u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) / 8388608.0f) - 1.0f; }

struct Buffers { uint16_t *A, *B; void* C; };
static bool g_bf16_out = false;

static void launch_f(const Buffers& d, int M, int N, int K, hipStream_t s) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (g_bf16_out) hipLaunchKernelGGL(gemm_f<true>, dim3(tiles), dim3(512), LDS_BYTES, s, d.A, d.B, d.C, M, N, K);
  else hipLaunchKernelGGL(gemm_f<false>, dim3(tiles), dim3(512), LDS_BYTES, s, d.A, d.B, d.C, M, N, K);
}
static int splits_for(int R, int I, int J, int& ktps) {
  const int tiles = ((I + BM - 1) / BM) * ((J + BN - 1) / BN), nk = (R + 63) / 64;
  int s = std::max(1, 256 / tiles);
  s = std::min(s, std::max(1, nk / 4));      // >= 4 k-tiles per split
  ktps = (nk + s - 1) / s;
  return (nk + ktps - 1) / ktps;
}
static void launch_w(const Buffers& d, int R, int I, int J, int splits, int ktps, hipStream_t s) {
  const int tiles = ((I + BM - 1) / BM) * ((J + BN - 1) / BN);
  hipLaunchKernelGGL(gemm_w, dim3(tiles, splits), dim3(512), LDS_BYTES, s, d.A, d.B, (float*)d.C, R, I, J, ktps);
}

static bool check_f(const Buffers& d, int M, int N, int K, uint64_t seed) {
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
  uint64_t s = seed;
  for (auto& v : ha) v = f2bf(frand(s));
  for (auto& v : hb) v = f2bf(frand(s) * 0.5f);
  CK(hipMemcpy(d.A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(d.C, 0xff, (size_t)M * N * 4));
  g_bf16_out = false;
  launch_f(d, M, N, K, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> hc((size_t)M * N);
  CK(hipMemcpy(hc.data(), d.C, hc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  const int ms = M <= 1024 ? 1 : 37, ns = N <= 1024 ? 1 : 29;      // every element for small cases, a row / column sample for large ones
  for (int m = 0; m < M; m += ms)
    for (int n = 0; n < N; n += ns) {
      double r = 0;
      for (int k = 0; k < K; ++k) r += (double)bf2f(ha[(size_t)m * K + k]) * bf2f(hb[(size_t)n * K + k]);
      const double e = std::fabs(r - hc[(size_t)m * N + n]);
      if (!(e <= worst)) worst = e;      // NaN-propagating max
    }
  const double bar = 2e-6 * K + 1e-5;
  printf("check F %5d x %5d x %5d vs fp64 on the same bf16 inputs: max abs err %.3e %s\n", M, N, K, worst, worst < bar ? "ok" : "WRONG");
  return worst < bar;
}
static bool check_w(const Buffers& d, int R, int I, int J, int splits_forced, uint64_t seed) {
  std::vector<uint16_t> ha((size_t)R * I), hb((size_t)R * J);
  uint64_t s = seed;
  for (auto& v : ha) v = f2bf(frand(s));
  for (auto& v : hb) v = f2bf(frand(s) * 0.5f);
  // what lies behind the last reduce row must not be read: poison it
  CK(hipMemset(d.A, 0x7f, ((size_t)R + 64) * I * 2));
  CK(hipMemset(d.B, 0x7f, ((size_t)R + 64) * J * 2));
  CK(hipMemcpy(d.A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  int ktps, splits = splits_for(R, I, J, ktps);
  if (splits_forced > 0) { const int nk = (R + 63) / 64; ktps = (nk + splits_forced - 1) / splits_forced; splits = (nk + ktps - 1) / ktps; }
  CK(hipMemset(d.C, 0xff, (size_t)splits * I * J * 4));
  launch_w(d, R, I, J, splits, ktps, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> hc((size_t)splits * I * J);
  CK(hipMemcpy(hc.data(), d.C, hc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  const int is = I <= 600 ? 1 : 23, js = J <= 600 ? 1 : 19;
  for (int i = 0; i < I; i += is)
    for (int j = 0; j < J; j += js) {
      double r = 0, got = 0;
      for (int k = 0; k < R; ++k) r += (double)bf2f(ha[(size_t)k * I + i]) * bf2f(hb[(size_t)k * J + j]);
      for (int z = 0; z < splits; ++z) got += hc[(size_t)z * I * J + (size_t)i * J + j];
      const double e = std::fabs(r - got);
      if (!(e <= worst)) worst = e;
    }
  const double bar = 2e-6 * R + 1e-5;
  printf("check W %5d rows, %5d x %5d, %2d splits vs fp64 on the same bf16 inputs: max abs err %.3e %s\n", R, I, J, splits, worst, worst < bar ? "ok" : "WRONG");
  return worst < bar;
}

int main(int argc, char** argv) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const size_t maxA = std::max((size_t)15424 * 2048, (size_t)8192 * 8192);
  const size_t maxB = maxA, maxC = std::max((size_t)5120 * 10547, (size_t)8192 * 8192);
  Buffers d;
  CK(hipMalloc(&d.A, maxA * 2)); CK(hipMalloc(&d.B, maxB * 2)); CK(hipMalloc(&d.C, maxC * 4));
  // ---- correctness: ragged extents, every tail length of the k-loop, several passes of a one-workgroup-per-CU case
  bool ok = true;
  ok &= check_f(d, 333, 261, 64, 1);
  ok &= check_f(d, 333, 261, 128, 2);
  ok &= check_f(d, 333, 261, 192, 3);
  ok &= check_f(d, 600, 515, 256, 4);
  ok &= check_f(d, 256, 256, 320, 5);
  ok &= check_f(d, 1100, 777, 512, 6);
  for (int rep = 0; rep < 2; ++rep) ok &= check_f(d, 4096, 4096, 2048, 7 + rep);
  ok &= check_w(d, 64, 256, 256, 1, 11);
  ok &= check_w(d, 100, 264, 520, 1, 12);
  ok &= check_w(d, 333, 512, 512, 1, 13);
  ok &= check_w(d, 333, 512, 512, 3, 14);
  ok &= check_w(d, 2977, 512, 2048, 0, 15);
  ok &= check_w(d, 2977, 10552, 512, 0, 16);
  for (int rep = 0; rep < 2; ++rep) ok &= check_w(d, 7680, 512, 2048, 0, 17 + rep);
  if (!ok) return 1;
  // ---- timing: per-launch HIP events, median of `reps`; uniform random [-1, 1) operands
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  {
    std::vector<uint16_t> h(std::max(maxA, maxB));
    uint64_t s = 99;
    for (auto& v : h) v = f2bf(frand(s));
    CK(hipMemcpy(d.A, h.data(), maxA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.B, h.data() + 12345, (maxB - 12345) * 2, hipMemcpyHostToDevice));
  }
  struct Shape { const char* name; int form, M, N, K; };      // W form: M = I, N = J, K = R
  const Shape shapes[] = {
    {"F ffn2  (NACF B=128)", 0, 5120, 512, 2048}, {"F enc_lin", 0, 7680, 512, 2048}, {"F kvmem", 0, 15360, 1024, 512},
    {"F vocab (NACF B=128)", 0, 5120, 10547, 512}, {"F vocab dX (2970 live rows)", 0, 2970, 512, 10560},
    {"W dW ffn (2970 live rows)", 1, 512, 2048, 2970}, {"W dW qkv (2970 live rows)", 1, 1536, 512, 2970}, {"W dW proj (2970 live rows)", 1, 512, 512, 2970},
    {"W dW vocab (2970 live rows)", 1, 10552, 512, 2970}, {"W dW enc_lin", 1, 512, 2048, 7680}, {"W dW enc_hw", 1, 1024, 512, 7680},
    {"W dW kvmem", 1, 1024, 512, 15360},
    {"F cube 4096", 0, 4096, 4096, 4096}, {"F cube 8192", 0, 8192, 8192, 8192}, {"W cube 4096", 1, 4096, 4096, 4096}, {"W cube 8192", 1, 8192, 8192, 8192},
  };
  std::vector<hipEvent_t> ev(2 * reps);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int pass = 0; pass < 2; ++pass) {
    g_bf16_out = pass == 1;
    printf("%-30s %-18s %10s %10s %8s   (256 x 256 x 64 eight-phase, F out %s, group_m %d)\n", "gemm", "M,N,K", "median us", "min us", "TF",
           g_bf16_out ? "bf16" : "fp32", GROUP_M);
    for (const Shape& sh : shapes) {
      if (pass == 1 && sh.form == 1) continue;
      int ktps = 0, splits = 1;
      if (sh.form == 1) splits = splits_for(sh.K, sh.M, sh.N, ktps);
      auto go = [&] { if (sh.form == 0) launch_f(d, sh.M, sh.N, sh.K, 0); else launch_w(d, sh.K, sh.M, sh.N, splits, ktps, 0); };
      for (int i = 0; i < 3; ++i) go();
      CK(hipDeviceSynchronize());
      for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(ev[2 * i], 0));
        go();
        CK(hipEventRecord(ev[2 * i + 1], 0));
      }
      CK(hipDeviceSynchronize());
      std::vector<float> ms(reps);
      for (int i = 0; i < reps; ++i) CK(hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]));
      std::sort(ms.begin(), ms.end());
      CK(hipEventRecord(ev[0], 0));
      for (int i = 0; i < reps; ++i) go();
      CK(hipEventRecord(ev[1], 0));
      CK(hipDeviceSynchronize());
      float chain_ms = 0.f;
      CK(hipEventElapsedTime(&chain_ms, ev[0], ev[1]));
      const double chain_us = chain_ms * 1e3 / reps;
      const double med = ms[reps / 2], flops = 2.0 * sh.M * sh.N * sh.K;
      const int wgs = ((sh.M + BM - 1) / BM) * ((sh.N + BN - 1) / BN) * splits;
      printf("%-30s %5d,%5d,%5d  %10.1f %10.1f %8.1f   %4d workgroups (%2d splits), %.3f of 2500 TF | back to back %7.1f us = %7.1f TF (%.3f)\n", sh.name,
             sh.M, sh.N, sh.K, med * 1e3, ms[0] * 1e3, flops / (med * 1e-3) / 1e12, wgs, splits, flops / (med * 1e-3) / 2.5e15, chain_us,
             flops / (chain_us * 1e-6) / 1e12, flops / (chain_us * 1e-6) / 2.5e15);
    }
  }
  return 0;
}

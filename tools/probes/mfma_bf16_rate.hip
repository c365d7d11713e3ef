// Sustained issue rate of the two bf16 MFMA shapes (tuning probe): independent accumulators, operands in registers.
// build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(int iters, float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f - i * 0.01f); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(int iters, float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f - i * 0.01f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  if (s == 123.456f) out[0] = s;
}
template <class K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int wgs) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, 100, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)wgs * 4 * iters * nacc * flop_per_mfma;
  printf("%-28s acc=%2d wgs=%4d  %.3f ms  %.0f TFLOP/s\n", name, nacc, wgs, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int wgs : {256, 512}) {
    run("v_mfma_f32_16x16x32_bf16", k16<4>, 4, 2.0 * 16 * 16 * 32, wgs);
    run("v_mfma_f32_16x16x32_bf16", k16<16>, 16, 2.0 * 16 * 16 * 32, wgs);
    run("v_mfma_f32_32x32x16_bf16", k32<2>, 2, 2.0 * 32 * 32 * 16, wgs);
    run("v_mfma_f32_32x32x16_bf16", k32<4>, 4, 2.0 * 32 * 32 * 16, wgs);
  }
  return 0;
}

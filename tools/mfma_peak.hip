// Attainable-peak probe: register-only v_mfma_f32_16x16x4_f32 loop (no memory traffic), to
// separate "the GEMM kernel leaves MFMA slots empty" from "the chip does not sustain the paper
// clock under matrix load".  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // RANDOM != 0: operands with full-entropy mantissas (switching activity of a real GEMM, what the power
  // management sees); RANDOM == 0: constant operands
  float a[4], b[4];
  unsigned h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    h = h * 1664525u + 1013904223u;
    a[r] = a0 != 0.f ? a0 : __uint_as_float(0x3f000000u | (h >> 9)) * ((h & 1) ? 1.f : -1.f);
    h = h * 1664525u + 1013904223u;
    b[r] = b0 != 0.f ? b0 : __uint_as_float(0x3f000000u | (h >> 9)) * ((h & 1) ? 1.f : -1.f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 3], acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

template <int NACC>
static void run(int blocks, int iters, float* d, float fill = 1.0f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters / 10, fill, fill);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, fill, fill);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * NACC * 2048.0;
  printf("acc=%2d blocks=%5d iters=%d %s operands  %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters,
         fill != 0.f ? "constant" : "random  ", ms, flops / ms * 1e-9);
}

int main(int argc, char** argv) {
  float* d;
  hipMalloc(&d, 256);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(256, iters, d);       // 1 wave / SIMD
    run<4>(512, iters, d);       // 2 waves / SIMD
    run<16>(256, iters, d);
    run<16>(512, iters, d);
    run<16>(1024, iters, d);
    run<16>(2048, iters / 2, d);
    run<16>(512, iters, d, 0.f);
    run<16>(1024, iters * 4, d, 0.f);   // long enough for the power controller to react
  }
  return 0;
}

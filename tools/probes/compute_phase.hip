// What does the MFMA phase of the exact 128x128 kernel cost by itself?  (tuning probe)  One workgroup of 4 waves per CU (or
// two), LDS images filled once, then per iteration exactly the phase's instruction stream: 12 + 4*3 fragment reads
// (ds_read_b128, the kernel's swizzled layout) and 96 v_mfma_f32_16x16x32_bf16 in the kernel's order (row block, term,
// column block), optionally a barrier.  Variants isolate the parts.   build: hipcc --offload-arch=gfx950 -O3 compute_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int lds_sw(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ bf16x8 frag(const u32x4* plane, int row, int g) { return __builtin_bit_cast(bf16x8, plane[row * 4 + (g ^ lds_sw(row))]); }

// MODE 0: reads + MFMAs + 2 barriers (the kernel's phase)   1: no barriers   2: fragments read once, MFMAs only
// 3: reads only (no MFMA; results kept alive)                 4: as 0 but term-outermost order over all 16 accumulators
template <int MODE>
__global__ __launch_bounds__(256, 2) void phase(int iters, float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 6 * 512; i += 256) smem[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const u32x4* qpl = smem;            // 3 planes x 512 chunks
  const u32x4* ppl = smem + 3 * 512;
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};
  bf16x8 pf[4][3], qf[4][3];
  if (MODE == 2) {
    for (int b = 0; b < 4; ++b) for (int p = 0; p < 3; ++p) pf[b][p] = frag(ppl + p * 512, wn * 64 + b * 16 + li, lg);
    for (int a = 0; a < 4; ++a) for (int p = 0; p < 3; ++p) qf[a][p] = frag(qpl + p * 512, wm * 64 + a * 16 + li, lg);
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int p = 0; p < 3; ++p) pf[b][p] = frag(ppl + p * 512, wn * 64 + b * 16 + li, lg);
    }
    if (MODE == 4) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) qf[a][p] = frag(qpl + p * 512, wm * 64 + a * 16 + li, lg);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][TP[t]], qf[a][TQ[t]], acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (MODE != 2) {
#pragma unroll
          for (int p = 0; p < 3; ++p) qf[a][p] = frag(qpl + p * 512, wm * 64 + a * 16 + li, lg);
        }
        if (MODE == 3) {
#pragma unroll
          for (int p = 0; p < 3; ++p) acc[a][p][0] += (float)qf[a][p][0] + (float)pf[a][p][1];
        } else {
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][TP[t]], qf[a][TQ[t]], acc[a][b], 0, 0, 0);
        }
      }
    }
    if (MODE == 0 || MODE == 4) { __syncthreads(); __syncthreads(); }
    if (MODE == 2 || MODE == 1 || MODE == 3) asm volatile("" ::: "memory");
  }
  const long long t1 = clock64();
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
  if (s == 123.456f) out[0] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int wgs) {
  float* out; long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(phase<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  const int iters = 2000;
  hipLaunchKernelGGL(phase<MODE>, dim3(wgs), dim3(256), 72 * 1024, 0, 50, out, cyc);
  hipLaunchKernelGGL(phase<MODE>, dim3(wgs), dim3(256), 72 * 1024, 0, iters, out, cyc);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, 8 * wgs, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < wgs; ++i) m += h[i]; m /= wgs;
  printf("%-58s %d per CU: %7.0f cycles per k-tile phase (96 MFMAs = 1536)\n", name, wgs / 256, m / iters);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int wgs : {256, 512}) {
    run<0>("reads + MFMAs (row block, term, column) + 2 barriers", wgs);
    run<1>("the same without barriers", wgs);
    run<2>("MFMAs only (fragments read once)", wgs);
    run<3>("fragment reads only", wgs);
    run<4>("reads + MFMAs term-outermost + 2 barriers", wgs);
  }
  return 0;
}

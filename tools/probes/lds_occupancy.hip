// How many 256-thread workgroups share a CU as a function of their dynamic LDS size and VGPR count?  (tuning probe)
// Every workgroup spins for a fixed number of shader clocks; the launch of 2 * CUs workgroups takes one spin if two fit
// per CU, two spins if only one does.  build: hipcc --offload-arch=gfx950 -O3 lds_occupancy.hip -o lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NV>
__global__ __launch_bounds__(256, 2) void spin(long long cycles, float* out, unsigned* cu_count) {
  extern __shared__ unsigned char smem[];
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.5f + i;
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + 0.5f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  if (s == 12345.f) { smem[threadIdx.x] = 1; out[0] = s + smem[(threadIdx.x + 1) & 255]; }
}

template <int NV>
void run(int lds_kb, int cus) {
  auto k = spin<NV>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, (size_t)lds_kb * 1024);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k));
  float* out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms[3];
  for (int mult = 1; mult <= 3; ++mult) {
    hipLaunchKernelGGL(k, dim3(cus * mult), dim3(256), lds_kb * 1024, 0, 200000LL, out, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(cus * mult), dim3(256), lds_kb * 1024, 0, 200000LL, out, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms[mult - 1], a, b);
  }
  printf("regs %3d  dynamic LDS %3d KB: occupancy API %d   time for 1x / 2x / 3x CUs workgroups: %.3f %.3f %.3f ms -> %s\n", fa.numRegs, lds_kb,
         occ, ms[0], ms[1], ms[2], ms[1] < 1.5f * ms[0] ? (ms[2] < 1.5f * ms[0] ? ">= 3 per CU" : "2 per CU") : "1 per CU");
  hipFree(out);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, regsPerBlock %d\n", p.gcnArchName, p.multiProcessorCount,
         (size_t)p.sharedMemPerMultiprocessor, (size_t)p.sharedMemPerBlock, p.regsPerBlock);
  for (int kb : {16, 32, 48, 53, 56, 64, 72, 80}) run<16>(kb, p.multiProcessorCount);
  for (int kb : {48, 64, 72, 80}) run<200>(kb, p.multiProcessorCount);
  return 0;
}

// The weight-gradient member of the wide-wave-tile GEMM family (gemm_bf16_widet.hpp): instantiation and grouped launch.
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE
#include "gemm_bf16_launch.hpp"
#include "gemm_bf16_widet.hpp"

void launch_widet_dw_group(const GemmGroup<EpiStore>& t, hipStream_t s) {
  auto kern = widet::gemm_widet_group_kernel;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, widet::LDS_BYTES);
    raised = true;
  }
  hipLaunchKernelGGL(kern, dim3(t.wg0[t.n]), dim3(256), widet::LDS_BYTES, s, t);
  bf16_note_wide("gemm_widet_group_kernel");       // (rocprofv3: widet::gemm_widet_group_kernel(GemmGroup<EpiStore>))
}

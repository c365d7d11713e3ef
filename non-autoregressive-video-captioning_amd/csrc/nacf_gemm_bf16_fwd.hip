// bf16 matrix-core instantiations: nn.Linear forward (fused EpiLinear epilogue)
#undef NACF_GEMM_TRACE
#include "gemm_bf16_launch.hpp"

static thread_local char g_last_kernel[160] = "";
const char* bf16_last_kernel_name() { return g_last_kernel; }
void bf16_note_kernel(int tile, int qsrc, int psrc, int ns, int stages, const char* epi) {
  snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_bf16_kernel<%d, %d, %d, %d, %d, %d, %s>", tile, tile, qsrc, psrc, ns,
           stages, epi);
}

void bf16_note_dma64(const char* epi) { snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_dma64_kernel<32, %s>", epi); }
void bf16_note_wide(const char* name) { snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", name); }

void launch_bf16_linear(GemmShape g, const EpiLinear& epi, int tile, int ns, hipStream_t s) {
  if (g.Pimg) launch_bf16_any<SRC_F32_KC, SRC_BF16_KC, EpiLinear>(g, epi, 1, tile, ns, s, "EpiLinear");
  else launch_bf16_any<SRC_F32_KC, SRC_F32_KC, EpiLinear>(g, epi, 1, tile, ns, s, "EpiLinear");
}

void launch_wimage_refresh(const WImageDesc* descs, int n_desc, int n_tiles, int ns, hipStream_t s) {
  if (ns == 1) hipLaunchKernelGGL(wimage_refresh_kernel<1>, dim3(n_tiles), dim3(256), 0, s, descs, n_desc);
  else hipLaunchKernelGGL(wimage_refresh_kernel<3>, dim3(n_tiles), dim3(256), 0, s, descs, n_desc);
}

#ifdef NACF_BF16_TRACE
extern "C" int nacf_debug_bf16_trace(void* buf) {     // tuning builds only (make trace); not part of the shipped ABI
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bf16_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

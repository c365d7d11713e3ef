// bf16 matrix-core instantiations: the two backward GEMMs of nn.Linear (EpiStore: accumulate / split-K slabs)
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE   // the phase stamps are compiled into the forward instantiations only
#include "gemm_bf16_launch.hpp"

void launch_bf16_dx(GemmShape g, const EpiStore& epi, int splits, int tile, int ns, hipStream_t s) {
  // P = W^T as a pre-split image (rows = output columns, k-contiguous), else the fp32 weight read row-contiguous
  if (g.Pimg) launch_bf16_any<SRC_F32_KC, SRC_BF16_KC, EpiStore>(g, epi, splits, tile, ns, s, "EpiStore");
  else launch_bf16_any<SRC_F32_KC, SRC_F32_MC, EpiStore>(g, epi, splits, tile, ns, s, "EpiStore");
}

void launch_bf16_dw(GemmShape g, const EpiStore& epi, int splits, int tile, int ns, hipStream_t s) {
  launch_bf16_any<SRC_F32_MC, SRC_F32_MC, EpiStore>(g, epi, splits, tile, ns, s, "EpiStore");
}

template <int NS, int STAGES>
static void launch_dw_group_one(const GemmGroup<EpiStore>& t, hipStream_t s) {
  constexpr size_t bytes = (size_t)gemm_bf16_lds_chunks<128, 128, NS, STAGES, false>() * 16;
  auto kern = gemm_bf16_group_kernel<128, 128, SRC_F32_MC, SRC_F32_MC, NS, STAGES, EpiStore>;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    raised = true;
  }
  hipLaunchKernelGGL(kern, dim3(t.wg0[t.n]), dim3(256), bytes, s, t);
}
void launch_bf16_dw_group(const GemmGroup<EpiStore>& t, int ns, hipStream_t s) {
  if (ns == 1) launch_dw_group_one<1, 2>(t, s);
  else launch_dw_group_one<3, NACF_BF16_EXACT128_STAGES>(t, s);
  char name[96];
  snprintf(name, sizeof(name), "gemm_bf16_group_kernel<128, 128, %d, %d, %d, %d, EpiStore>", SRC_F32_MC, SRC_F32_MC, ns == 1 ? 1 : 3,
           ns == 1 ? 2 : NACF_BF16_EXACT128_STAGES);
  bf16_note_wide(name);        // (sets the "last kernel" string verbatim: rocprofv3's name of the grouped kernel)
}

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

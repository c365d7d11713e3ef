// bf16 matrix-core instantiations: vocabulary projection with the fused soft-max statistics (EpiArgmax)
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE   // the phase stamps are compiled into the forward instantiations only
#include "gemm_bf16_launch.hpp"

void launch_bf16_argmax(GemmShape g, const EpiArgmax& epi, int tile, int ns, hipStream_t s) {
  if (g.Pimg) launch_bf16_any<SRC_F32_KC, SRC_BF16_KC, EpiArgmax>(g, epi, 1, tile, ns, s, "EpiArgmax");
  else launch_bf16_any<SRC_F32_KC, SRC_F32_KC, EpiArgmax>(g, epi, 1, tile, ns, s, "EpiArgmax");
}

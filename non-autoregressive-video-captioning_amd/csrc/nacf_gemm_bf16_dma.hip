// The DMA-staged throughput-mode GEMM (gemm_bf16_dma.hpp): instantiations and the choice between it and the register-staged
// 64x64 / 128x128 kernels of gemm_bf16.hpp.
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE
#include "gemm_bf16_launch.hpp"
#include "gemm_bf16_dma.hpp"

namespace {
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// NACF_GEMM_DMA: 1 = whenever the launch is eligible, otherwise never.  Read per call (tests switch it); NACF_GEMM_TILE (the
// 64 / 128 knob of the other kernels) also turns this kernel off.
int dma_env() { const char* e = getenv("NACF_GEMM_DMA"); return e ? atoi(e) : -1; }

template <class Epi>
void launch_dma_one(GemmShape g, const Epi& epi, int splits, const char* name, hipStream_t s) {
  auto kern = dma::gemm_dma_kernel<Epi>;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dma::LDS_BYTES);
    raised = true;
  }
  g.tiles_m = cdiv(g.M, dma::BM);
  g.tiles_n = cdiv(g.N, dma::BN);
  g.group_n = (g.tiles_n >= 32 && splits == 1) ? 6 : 0;       // L2-aware order for vocabulary-wide P (GemmShape::group_n)
  dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), dma::LDS_BYTES, s, g, epi);
  bf16_note_wide(name);
}
}  // namespace

// Eligible: throughput mode with a one-plane weight image, whole 32-wide k-tiles per reduce split, 16-byte addressable
// activations.  OPT-IN (NACF_GEMM_DMA=1): measured on the NACF step's shapes (tools/gemm_bench.py --modes bf16 --tiles
// 64,128,dma --images; profiles/r03_dma_gemm_bench.txt) it ties or wins on the dX launches with a long reduce dimension
// (7680x512 <- 1024: 346 vs 334 TF, 15360x512 <- 1024: 491 vs 518) and loses on the K = 512 forward launches (242 vs 369 TF):
// with fp32 activations a 128 x 128 tile asks the CU's global -> LDS path for 24 KB per 256 matrix-pipe cycles (96 B / clk
// against the ~52 B / clk it delivers: one 1 KB request per ~20 cycles), so the kernel is feed-bound where the register-staged
// kernels are staging-bound.  It is the kernel for bf16-RESIDENT activations (half the Q bytes, no conversions:
// tools/probes/bf16_resident_gemm.hip, 12 us for 5120x512x512 where the register-staged kernels take 22) -- kept behind the
// knob, with its tests, until the producers write bf16 copies.
bool dma_pick(const GemmShape& g, int splits, bool has_rows, int ns, bool heavy_epilogue) {
  const int forced = dma_env();
  if (forced != 1 || ns != 1 || !g.Pimg || getenv("NACF_GEMM_TILE")) return false;
  const int kps = splits > 1 ? g.k_per_split : g.K;
  if (g.K % dma::BK != 0 || kps % dma::BK != 0 || kps < dma::BK || (splits > 1 && g.K % kps != 0)) return false;
  if (!al16(g.Q) || g.ldq % 4 != 0) return false;
  (void)has_rows; (void)heavy_epilogue;
  return true;
}
bool launch_dma_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, bool heavy, hipStream_t s) {
  if (!dma_pick(g, 1, has_rows, 1, heavy)) return false;
  launch_dma_one<EpiLinear>(g, epi, 1, "gemm_dma_kernel<EpiLinear>", s);
  return true;
}
bool launch_dma_dx(const GemmShape& g, const EpiStore& epi, int splits, bool has_rows, hipStream_t s) {
  if (!dma_pick(g, splits, has_rows, 1, false)) return false;
  launch_dma_one<EpiStore>(g, epi, splits, "gemm_dma_kernel<EpiStore>", s);
  return true;
}

// Host-side launchers of the bf16 matrix-core GEMM family (gemm_bf16.hpp).  The kernels are instantiated in three
// translation units (nacf_gemm_bf16_{fwd,vocab,bwd}.hip) so that the library builds in parallel.
//   tile: 0 = 128x128, 1 = 64x64 workgroup tile;  ns: 1 = bf16 throughput, 3 = exact fp32 split (six MFMAs per block)
//   g.tiles_m / g.tiles_n are filled in by the launcher; g.Pimg != nullptr selects the pre-split weight image as P.
#pragma once
#include <cstdlib>
#include "gemm_bf16.hpp"
#include "gemm_dma64.hpp"

#ifndef NACF_BF16_EXACT128_STAGES
#define NACF_BF16_EXACT128_STAGES 3     // LDS staging of the exact mode's 128x128 tile: 1 | 3 (gemm_bf16.hpp "STAGES")
#endif

void launch_bf16_linear(GemmShape g, const EpiLinear& epi, int tile, int ns, hipStream_t s);              // y = x W^T
void launch_bf16_argmax(GemmShape g, const EpiArgmax& epi, int tile, int ns, hipStream_t s);              // + soft-max stats
void launch_bf16_dx(GemmShape g, const EpiStore& epi, int splits, int tile, int ns, hipStream_t s);       // dX = dZ W
void launch_bf16_dw(GemmShape g, const EpiStore& epi, int splits, int tile, int ns, hipStream_t s);       // dW = dZ^T X
void launch_bf16_dw_group(const GemmGroup<EpiStore>& t, int ns, hipStream_t s);
// wide-wave-tile kernels (gemm_bf16_wide.hpp, nacf_gemm_bf16_wide.hip): false = not eligible / not worthwhile, use the others
bool launch_wide_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, bool heavy_epilogue, hipStream_t s);
bool launch_wide_dx(const GemmShape& g, const EpiStore& epi, int splits, bool has_rows, hipStream_t s);
int wide_pick(const GemmShape& g, int splits, bool has_rows, int ns, bool heavy_epilogue);
// DMA-fed two-per-CU kernels of the exact mode (gemm_dma128.hpp, nacf_gemm_dma128.hip): false / 0 = not eligible, use the others
int dma128_pick(const GemmShape& g, int splits, bool has_rows, int ns, int kind);
bool launch_dma128_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, hipStream_t s);
bool launch_dma128_dx(const GemmShape& g, const EpiStore& epi, int splits, bool has_rows, hipStream_t s);
void launch_dma128_argmax(const GemmShape& g, const EpiArgmax& epi, int mt, hipStream_t s);
void bf16_note_wide(const char* name);
void launch_wimage_refresh(const WImageDesc* descs, int n_desc, int n_tiles, int ns, hipStream_t s);
// "gemm_bf16_kernel<BM, BN, QSRC, PSRC, NS, STAGES, Epi>" of the launch the calling thread made last (profiling aid)
const char* bf16_last_kernel_name();
void bf16_note_kernel(int tile, int qsrc, int psrc, int ns, int stages, const char* epi);
void bf16_note_dma64(const char* epi);

template <int BM, int QSRC, int PSRC, int NS, int STAGES, class Epi>
inline void launch_bf16_one(const GemmShape& g, const Epi& epi, dim3 grid, hipStream_t s) {
  constexpr size_t bytes = (size_t)gemm_bf16_lds_chunks<BM, BM, NS, STAGES, Epi::kArgmax>() * 16;
  auto kern = gemm_bf16_kernel<BM, BM, QSRC, PSRC, NS, STAGES, Epi>;
  if constexpr (bytes > 48 * 1024) {          // above the default dynamic-LDS limit: raise it once per kernel
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      raised = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), bytes, s, g, epi);
}

template <int QSRC, int PSRC, class Epi>
inline void launch_bf16_any(GemmShape g, const Epi& epi, int splits, int tile, int ns, hipStream_t s, const char* epi_name) {
  const int t = tile == 0 ? 128 : 64;
  g.tiles_m = cdiv(g.M, t);
  g.tiles_n = cdiv(g.N, t);
  dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
  {
    // L2-aware order for very wide P (GemmShape::group_n): measured on the vocabulary projection (83 column tiles),
    // groups of 4..16 all give +5-6 % (exact) / +11 % (bf16) over n-fastest; NACF_GEMM_GROUP_N overrides (0 = off)
    static const int group_n = [] { const char* e = getenv("NACF_GEMM_GROUP_N"); return e ? atoi(e) : 6; }();
    if (group_n > 0 && g.tiles_n >= 32 && splits == 1) g.group_n = group_n;
  }
  // throughput mode, the 64 x 64 tile with a weight image, launches of at most 3 tiles per CU: the DMA-fed kernel of
  // gemm_dma64.hpp on 32 x 64 tiles (NACF_DMA64=0: gemm_bf16_kernel<64, 64>).  Threshold scan on the NACF / NAB steps:
  // 768 / 1024 / 2048 tiles -> 1.694 / 1.720 / 1.725 ms (NACF B = 128), 1.019 / 1.023 / 1.024 ms (NAB B = 64).
  if constexpr (QSRC == SRC_F32_KC && PSRC == SRC_BF16_KC) {
    static const int max_tiles = [] { const char* e = getenv("NACF_DMA64_MAX"); return e ? atoi(e) : 768; }();
    if (ns == 1 && tile != 0 && g.tiles_m * g.tiles_n * splits <= max_tiles && dma64::eligible(g, splits)) {
      const char* e = getenv("NACF_DMA64");
      if (!e || atoi(e) != 0) {
        g.tiles_m = cdiv(g.M, 32);
        grid.x = (g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n;
        dma64::launch<32, Epi>(g, epi, grid, s);
        bf16_note_dma64(epi_name);
        return;
      }
    }
  }
  // LDS images: two of each operand where they fit (one barrier per k-tile); the exact mode's 128x128 tile keeps one
  // image of Q and two of P (72 KB, two workgroups per CU): P is loaded during the MFMA phase, two tiles ahead
  if (ns == 1) {
    if (tile == 0) launch_bf16_one<128, QSRC, PSRC, 1, 2, Epi>(g, epi, grid, s);
    else launch_bf16_one<64, QSRC, PSRC, 1, 2, Epi>(g, epi, grid, s);
    bf16_note_kernel(t, QSRC, PSRC, 1, 2, epi_name);
  } else {
    if (tile == 0) launch_bf16_one<128, QSRC, PSRC, 3, NACF_BF16_EXACT128_STAGES, Epi>(g, epi, grid, s);
    else launch_bf16_one<64, QSRC, PSRC, 3, 2, Epi>(g, epi, grid, s);
    bf16_note_kernel(t, QSRC, PSRC, 3, tile == 0 ? NACF_BF16_EXACT128_STAGES : 2, epi_name);
  }
}

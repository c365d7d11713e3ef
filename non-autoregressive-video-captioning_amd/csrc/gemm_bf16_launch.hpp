// Host-side launchers of the bf16 matrix-core GEMM family (gemm_bf16.hpp).  The kernels are instantiated in three
// translation units (nacf_gemm_bf16_{fwd,vocab,bwd}.hip) so that the library builds in parallel.
//   tile: 0 = 128x128, 1 = 64x64 workgroup tile;  ns: 1 = bf16 throughput, 3 = exact fp32 split (six MFMAs per block)
//   g.tiles_m / g.tiles_n are filled in by the launcher; g.Pimg != nullptr selects the pre-split weight image as P.
#pragma once
#include "gemm_bf16.hpp"

void launch_bf16_linear(GemmShape g, const EpiLinear& epi, int tile, int ns, hipStream_t s);              // y = x W^T
void launch_bf16_argmax(GemmShape g, const EpiArgmax& epi, int tile, int ns, hipStream_t s);              // + soft-max stats
void launch_bf16_dx(GemmShape g, const EpiStore& epi, int splits, int tile, int ns, hipStream_t s);       // dX = dZ W
void launch_bf16_dw(GemmShape g, const EpiStore& epi, int splits, int tile, int ns, hipStream_t s);       // dW = dZ^T X
void launch_wimage_refresh(const WImageDesc* descs, int n_desc, int n_tiles, int ns, hipStream_t s);
// "gemm_bf16_kernel<BM, BN, QSRC, PSRC, NS, STAGES, Epi>" of the launch the calling thread made last (profiling aid)
const char* bf16_last_kernel_name();
void bf16_note_kernel(int tile, int qsrc, int psrc, int ns, int stages, const char* epi);

template <int QSRC, int PSRC, class Epi>
inline void launch_bf16_any(GemmShape g, const Epi& epi, int splits, int tile, int ns, hipStream_t s, const char* epi_name) {
  const int t = tile == 0 ? 128 : 64;
  g.tiles_m = cdiv(g.M, t);
  g.tiles_n = cdiv(g.N, t);
  dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
  // LDS images: two where they fit in 48 KB (one barrier per k-tile), one for the exact mode's 128x128 tile
  if (ns == 1) {
    if (tile == 0) hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, QSRC, PSRC, 1, 2, Epi>), grid, dim3(256), 0, s, g, epi);
    else hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, QSRC, PSRC, 1, 2, Epi>), grid, dim3(256), 0, s, g, epi);
    bf16_note_kernel(t, QSRC, PSRC, 1, 2, epi_name);
  } else {
    if (tile == 0) hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, QSRC, PSRC, 3, 1, Epi>), grid, dim3(256), 0, s, g, epi);
    else hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, QSRC, PSRC, 3, 2, Epi>), grid, dim3(256), 0, s, g, epi);
    bf16_note_kernel(t, QSRC, PSRC, 3, tile == 0 ? 1 : 2, epi_name);
  }
}

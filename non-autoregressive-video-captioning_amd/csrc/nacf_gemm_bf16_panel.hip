// The panel member of the bf16 matrix-core GEMM family (gemm_bf16_panel.hpp): instantiations and the choice between it and
// the 64x64 / 128x128 / wide kernels.
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE
#include "gemm_bf16_launch.hpp"
#include "gemm_bf16_panel.hpp"

namespace {
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// NACF_GEMM_PANEL: unset / 0 = never (the default: see panel_pick), 1 = whenever the launch is eligible, 2 = by shape.
// Read per call (tests switch it); NACF_GEMM_TILE / NACF_GEMM_WIDE (the knobs of the other kernels) also turn the panel kernel off.
int panel_env() { const char* e = getenv("NACF_GEMM_PANEL"); return e ? atoi(e) : 0; }

int n_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

// Eligible: exact mode, a fragment-major image, N a multiple of 128, the reduce dimension a multiple of 256 from 512 on,
// 16-byte addressable activations whose extent fits 32-bit byte offsets.
// Worthwhile: only the decoder layer's launches with a LONG reduce dimension (FFN2 forward, FFN1 dX: 2048; q|k|v dX: 1536).
//   tools/probes/panel_gemm.hip (back-to-back launches, plain epilogue, old 64x64 -> panel, live rows / slots):
//     reduce dimension 2048:  2552 / 5120 x 512   55.9 -> 40.0 us      3822 / 14592 x 512    65.3 -> 52.7
//     reduce dimension 1536:  2556 / 5120 x 512   43.2 -> 33.0
//     reduce dimension  512:  2657 / 5120 x 2048  47.8 -> 44.9         3714 / 14592 x 2048   75.5 -> 68.9
//                             2505 / 5120 x 1536  34.3 -> 36.4         8860 / 14592 x 512    39.4 -> 46.6
//                             2613 / 5120 x 512   17.9 -> 16.8          738 / 1280  x 512    11.1 -> 14.4
//   tools/chain_probe2.py (a graph of 20 launches, the layer's real epilogues, 2925 live rows): FFN2 57.0 -> 51.4 us, but
//     q|k|v 44.9 -> 46.2, output projection (dropout + residual) 20.9 -> 27.3, FFN1 (gelu) 54.2 -> 59.8: at K = 512 a tile is
//     8 k-steps per wave between a prologue and an epilogue of the same length, and the epilogue (partial sums through LDS,
//     then the fused nn.Linear tail on a quarter of the threads' registers) is slower than the 64x64 kernel's.
//   A first heuristic that also took the encoder's launches (K = 2048 without a row list) away from the grouped wide kernel cost
//   the step 0.12 ms (profiles/r04_panel_ab.txt); the shape rule below (NACF_GEMM_PANEL=2) is +-0 on the NACF step (2.618 vs 2.618 ms,
//   three alternating runs in one box) and -1 % on NA decode (4.94 vs 4.89 ms per batch): isolated launches gain 5-15 us, in the
//   step's graph they do not.  So the kernel is OFF by default; it stays as the body of the layer chain (gemm_bf16_chain.hpp).
bool panel_pick(const GemmShape& g, bool has_rows) {
  const int forced = panel_env();
  if (forced == 0 || getenv("NACF_GEMM_TILE") || getenv("NACF_GEMM_WIDE")) return false;
  if (!panel_eligible(g)) return false;
  if (forced == 1) return true;
  if (!has_rows || g.K < 1536) return false;
  const int m_eff = (int)((long)g.M * 29 / 50);                         // ~58 % of the slots are live (not known to the host)
  const long tiles = (long)cdiv(m_eff > 0 ? m_eff : 1, 64) * (g.N / 128);
  return tiles >= n_cus() / 4 && tiles <= 4L * n_cus();
}

}  // namespace

// what the kernel body can run at all (gemm_bf16_launch.hpp: also the chain kernel's test)
bool panel_eligible(const GemmShape& g) {
  if (!g.Pfrag || g.N % 128 != 0 || g.K % 256 != 0 || g.K < 512 || g.k_per_split != g.K) return false;
  return al16(g.Q) && g.ldq % 4 == 0 && (int64_t)g.M * g.ldq * 4 < (1LL << 32);
}

namespace {
template <class Epi>
bool launch_panel_any(const GemmShape& g0, const Epi& epi, bool has_rows, const char* name, hipStream_t s) {
  if (!panel_pick(g0, has_rows)) return false;
  using G = panel::Geo<2, 4, 3>;
  auto kern = panel::gemm_panel_kernel<2, 4, 3, Epi>;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    raised = true;
  }
  GemmShape g = g0;
  g.tiles_m = cdiv(g.M, G::BM);
  g.tiles_n = g.N / G::BN;
  const long tiles = (long)g.tiles_m * g.tiles_n;
  const int grid = tiles >= n_cus() ? n_cus() : (int)((tiles + 7) / 8 * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, s, g, epi);
  char buf[96];
  snprintf(buf, sizeof(buf), "gemm_panel_kernel<2, 4, 3, %s>", name);
  bf16_note_wide(buf);
  return true;
}
}  // namespace

bool launch_panel_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, hipStream_t s) {
  return launch_panel_any<EpiLinear>(g, epi, has_rows, "EpiLinear", s);
}
bool launch_panel_dx(const GemmShape& g, const EpiStore& epi, bool has_rows, hipStream_t s) {
  return launch_panel_any<EpiStore>(g, epi, has_rows, "EpiStore", s);
}

// The DMA-fed two-per-CU member of the exact mode's forward / dX GEMMs (gemm_dma128.hpp): instantiations and the choice between
// it, the wide kernels and the register-staged 128 x 128 / 64 x 64 kernels.
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE
#include "gemm_bf16_launch.hpp"
#include "gemm_dma128.hpp"

namespace {
// NACF_DMA128: 0 = never, 1 / 2 = that row-block count (64 / 128-row tiles) whenever the launch is eligible, unset = by size.
// Read per call (tests switch it); NACF_GEMM_TILE (the 64 / 128 knob of the register-staged kernels) also turns it off.
int dma128_env() { const char* e = getenv("NACF_DMA128"); return e ? atoi(e) : -1; }

thread_local char g_name[96] = "";
void note(int mt, const char* epi) {
  snprintf(g_name, sizeof(g_name), "gemm_dma128_kernel<%d, %s>", mt, epi);
  bf16_note_wide(g_name);
}
}  // namespace

// 0 = use the other kernels, 1 / 2 = this kernel with 64- / 128-row tiles.
int dma128_pick(const GemmShape& g, int splits, bool has_rows, int ns) {
  const int forced = dma128_env();
  if (forced == 0 || ns != 3 || getenv("NACF_GEMM_TILE") || !dma128::eligible(g, splits)) return 0;
  if (forced == 1 || forced == 2) return forced;
  // the 128-row tile when the launch has about a round of them (512 resident workgroups), else the 64-row tile; ~58 % of the
  // slots of a row list are live (not known to the host)
  const int m_eff = has_rows ? (int)((long)g.M * 29 / 50) : g.M;
  const long t2 = (long)cdiv(m_eff > 0 ? m_eff : 1, 128) * cdiv(g.N, dma128::BN) * splits;
  return t2 >= 384 ? 2 : 1;
}

bool launch_dma128_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, hipStream_t s) {
  const int mt = dma128_pick(g, 1, has_rows, 3);
  if (!mt) return false;
  if (mt == 2) dma128::launch<2, EpiLinear>(g, epi, 1, s); else dma128::launch<1, EpiLinear>(g, epi, 1, s);
  note(mt, "EpiLinear");
  return true;
}
bool launch_dma128_dx(const GemmShape& g, const EpiStore& epi, int splits, bool has_rows, hipStream_t s) {
  const int mt = dma128_pick(g, splits, has_rows, 3);
  if (!mt) return false;
  if (mt == 2) dma128::launch<2, EpiStore>(g, epi, splits, s); else dma128::launch<1, EpiStore>(g, epi, splits, s);
  note(mt, "EpiStore");
  return true;
}
// the vocabulary projection with soft-max statistics: the caller lays its partials out for 128-column tiles when `mt` != 0
void launch_dma128_argmax(const GemmShape& g, const EpiArgmax& epi, int mt, hipStream_t s) {
  if (mt == 2) dma128::launch<2, EpiArgmax>(g, epi, 1, s); else dma128::launch<1, EpiArgmax>(g, epi, 1, s);
  note(mt, "EpiArgmax");
}

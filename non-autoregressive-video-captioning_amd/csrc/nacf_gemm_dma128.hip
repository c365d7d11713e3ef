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

// 0 = use the other kernels, 1 / 2 = this kernel with 64- / 128-row tiles.  kind: 0 = nn.Linear forward (EpiLinear), 1 = dX / slabs
// (EpiStore), 2 = vocabulary projection with soft-max statistics (EpiArgmax).
// Measured inside the captured NACF step, the NA decode loop and tools/probes/dma128_probe.hip (profiles/r06_dma128_*):
//   * plain-store and soft-max epilogues, launches of a round of tiles or more: -2 ... -12 % per launch (vocabulary dX 156-165 ->
//     146-147 us, FFN dX 44 -> 39 and 35 -> 31, the decode's vocabulary projection 228 -> 221-223 us), a reduce dimension of 2048
//     -20 %; launches of under ~250 tiles lose 1-5 us.  On the whole step and the whole decode batch this is inside the run-to-run
//     spread (2.547 vs 2.549 ms, 4.71 vs 4.75 ms, three alternating runs).
//   * nn.Linear epilogues: the instantiation carries the family's whole activation / dropout switch (92 k instructions); hipcc then
//     runs the kernel at 256 registers with 300+ spill slots outside the steady loop (the straight-line tail k-tiles and the
//     epilogue), and even the dropout-free launches of inference lose (83 -> 110 us per 128-row launch, decode 4.7 -> 5.15 ms per batch);
//     with Philox dropout / gelu (20-80 k cycles per workgroup, which want the three small workgroups per CU of the 64 x 64 kernel
//     to hide behind) they lose 30-90 %.  nn.Linear forward therefore stays on the other kernels (NACF_DMA128_LINEAR=1 | 2 to A/B).
int dma128_pick(const GemmShape& g, int splits, bool has_rows, int ns, int kind) {
  const int forced = dma128_env();
  if (forced == 0 || ns != 3 || getenv("NACF_GEMM_TILE") || !dma128::eligible(g, splits)) return 0;
  if (forced == 1 || forced == 2) return forced;
  // kind 0 comes in two flavours: 0 = an epilogue with dropout (never), 3 = without (inference, q|k|v): from a round of 128-row tiles on
  const char* lin = getenv("NACF_DMA128_LINEAR");      // 0 = no nn.Linear forward here (default), 1 = the dropout-free ones from a round of 128-row tiles on, 2 = all of them by size
  const int lin_mode = lin ? atoi(lin) : 0;
  if ((kind == 0 && lin_mode < 2) || (kind == 3 && lin_mode < 1)) return 0;
  // ~58 % of the slots of a row list are live (not known to the host)
  const int m_eff = has_rows ? (int)((long)g.M * 29 / 50) : g.M;
  const long tn = cdiv(g.N, dma128::BN);
  const long t1 = (long)cdiv(m_eff > 0 ? m_eff : 1, 64) * tn * splits, t2 = (long)cdiv(m_eff > 0 ? m_eff : 1, 128) * tn * splits;
  static const int min_tiles = [] { const char* e = getenv("NACF_DMA128_MIN"); return e ? atoi(e) : 256; }();
  if (t1 < min_tiles) return 0;
  if (kind == 3 && lin_mode == 1 && t2 < 384) return 0;
  return t2 >= 384 ? 2 : 1;      // the 128-row tile from about a round of them on (512 resident workgroups)
}

bool launch_dma128_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, hipStream_t s) {
  const bool drop = epi.ep.p_drop1 > 0.f || epi.ep.p_drop2 > 0.f;
  const int mt = dma128_pick(g, 1, has_rows, 3, drop ? 0 : 3);
  if (!mt) return false;
  if (mt == 2) dma128::launch<2, EpiLinear>(g, epi, 1, s); else dma128::launch<1, EpiLinear>(g, epi, 1, s);
  note(mt, "EpiLinear");
  return true;
}
bool launch_dma128_dx(const GemmShape& g, const EpiStore& epi, int splits, bool has_rows, hipStream_t s) {
  const int mt = dma128_pick(g, splits, has_rows, 3, 1);
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

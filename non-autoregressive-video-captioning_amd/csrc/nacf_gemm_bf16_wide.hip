// The wide-wave-tile member of the bf16 matrix-core GEMM family (gemm_bf16_wide.hpp): instantiations, the choice between
// it and the 128x128 / 64x64 kernels, and the grouped launch of independent problems (the two encoder streams).
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE
#define NACF_PHILOX_MULHI 1      // common.hpp: this unit keeps the mul_hi / mul_lo form of the Philox round
#include <vector>
#include <mutex>
#include "gemm_bf16_launch.hpp"
#include "gemm_bf16_wide.hpp"

namespace {
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// NACF_GEMM_WIDE: 0 = never, 1 / 2 = that row-block count whenever the launch is eligible, unset = by size.
// Read per call (tests switch it); NACF_GEMM_TILE (the 64 / 128 knob of the other kernels) also turns the wide kernel off.
int wide_env() { const char* e = getenv("NACF_GEMM_WIDE"); return e ? atoi(e) : -1; }

template <int MT, class Epi>
void launch_wide_one(GemmShape g, const Epi& epi, int splits, hipStream_t s) {
  using G = wide::Geo<MT>;
  auto kern = wide::gemm_wide_kernel<MT, Epi>;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    raised = true;
  }
  g.tiles_m = cdiv(g.M, G::BM);
  g.tiles_n = cdiv(g.N, wide::BN);
  g.group_n = (g.tiles_n >= 16 && splits == 1) ? 3 : 0;       // L2-aware order for vocabulary-wide P (GemmShape::group_n)
  dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, s, g, epi);
}

// ---- grouped launch: independent problems of one epilogue type in a single grid (see gemm_bf16_group_kernel)
constexpr int WIDE_GROUP_MAX = 4;
template <class Epi>
struct WideGroup {
  int n;
  int wg0[WIDE_GROUP_MAX + 1];
  GemmShape g[WIDE_GROUP_MAX];
  Epi e[WIDE_GROUP_MAX];
};
template <int MT, class Epi>
__global__ __launch_bounds__(256, 1) void gemm_wide_group_kernel(WideGroup<Epi> t) {
  int p = 0;
#pragma unroll 1
  while (p + 1 < t.n && (int)blockIdx.x >= t.wg0[p + 1]) ++p;
  wide::gemm_wide_body<MT, Epi, 0>(t.g[p], t.e[p], (int)blockIdx.x - t.wg0[p], 0, 1);
}
template <int MT, class Epi>
void launch_wide_group(const std::vector<std::pair<GemmShape, Epi>>& items, hipStream_t s) {
  using G = wide::Geo<MT>;
  auto kern = gemm_wide_group_kernel<MT, Epi>;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    raised = true;
  }
  for (size_t i0 = 0; i0 < items.size(); i0 += WIDE_GROUP_MAX) {
    WideGroup<Epi> t;
    t.n = 0;
    int wg = 0;
    for (size_t i = i0; i < items.size() && t.n < WIDE_GROUP_MAX; ++i) {
      GemmShape g = items[i].first;
      g.tiles_m = cdiv(g.M, G::BM);
      g.tiles_n = cdiv(g.N, wide::BN);
      g.group_n = 0;
      const int gx = ((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n + 7) / 8 * 8;    // whole XCD rounds: blockIdx & 7 stays the XCD
      t.wg0[t.n] = wg;
      t.g[t.n] = g;
      t.e[t.n] = items[i].second;
      wg += gx;
      ++t.n;
    }
    t.wg0[t.n] = wg;
    hipLaunchKernelGGL(kern, dim3(wg), dim3(256), G::LDS_BYTES, s, t);
  }
}

// The group queue is PER HOST THREAD: a `with ops.wide_group()` block opens, fills and flushes it from one
// Python function, i.e. one thread -- the forward on the caller's, the HighWay dX pair on autograd's -- and a GEMM issued
// meanwhile by any other thread (another stream's work) launches at once instead of landing in somebody else's group.
thread_local bool g_wide_group_on = false;
thread_local std::vector<std::pair<GemmShape, EpiLinear>> g_wide_lin;
thread_local std::vector<std::pair<GemmShape, EpiStore>> g_wide_sto;
thread_local char g_wide_last[96] = "";
void note(int mt, const char* epi, bool grouped) {
  snprintf(g_wide_last, sizeof(g_wide_last), "gemm_wide_%skernel<%d, %s>", grouped ? "group_" : "", mt, epi);
  bf16_note_wide(g_wide_last);
}
long tiles_of(const GemmShape& g, int bm, int splits, bool has_rows) {
  const int m_eff = has_rows ? (int)((long)g.M * 29 / 50) : g.M;      // ~58 % of the slots are live (not known to the host)
  return (long)cdiv(m_eff > 0 ? m_eff : 1, bm) * cdiv(g.N, wide::BN) * splits;
}
template <class Epi>
int flush_one(std::vector<std::pair<GemmShape, Epi>>& v, const char* name, hipStream_t s) {
  if (v.empty()) return 0;
  long t2 = 0;
  for (auto& it : v) t2 += tiles_of(it.first, 128, 1, it.first.rows != nullptr);
  const int forced = wide_env();
  const int mt = (forced == 1 || forced == 2) ? forced : (t2 >= 160 ? 2 : 1);
  if (mt == 2) launch_wide_group<2, Epi>(v, s); else launch_wide_group<1, Epi>(v, s);
  note(mt, name, true);
  const int n = (int)v.size();
  v.clear();
  return n;
}
}  // namespace

// 0 = use the other kernels, 1 / 2 = the wide kernel with that many 32-row blocks per wave.
// Eligible: exact mode with a pre-split weight image, whole k-tiles in an even number >= 4 per reduce split (K % 64 == 0,
// K >= 128), 16-byte addressable activations whose extent fits 32-bit byte offsets.  Worthwhile (tools/probes/wide_gemm.hip,
// 1x MI355X, exact mode, old -> wide): 7680x512x2048 105 -> 91 us with 64-row tiles (120 big tiles: the 128-row tile fills
// half the chip, 120 us); 2688x2048x512 48 -> 42; 7680x1024x512 53 -> 50; 15360x1024x512 97 -> 94; 15360x512x1024 94 -> 87;
// the vocabulary (756 tiles, K = 512) ties (and has no wide soft-max epilogue instantiated).  One workgroup per CU means nothing overlaps a workgroup's epilogue, so
// transcendental / dropout epilogues stay with the 2-per-CU kernels unless the reduce dimension is long.
int wide_pick(const GemmShape& g, int splits, bool has_rows, int ns, bool heavy_epilogue) {
  const int forced = wide_env();
  if (forced == 0 || ns != 3 || !g.Pimg || getenv("NACF_GEMM_TILE")) return 0;
  const int kps = splits > 1 ? g.k_per_split : g.K;
  if (g.K % 64 != 0 || kps % 64 != 0 || kps < 128 || (splits > 1 && g.K % kps != 0)) return 0;
  if (!al16(g.Q) || g.ldq % 4 != 0 || (int64_t)g.M * g.ldq * 4 >= (1LL << 32) || (int64_t)g.N * 64 >= (1LL << 32)) return 0;
  if (forced == 1 || forced == 2) return forced;
  // (round 4 A/B, one box: heavy epilogues on this kernel from K = 512 on -- HighWay forward, the decoder's dropout
  //  epilogues -- 2.78 -> 2.86 ms per step)
  if (heavy_epilogue && kps < 2048) return 0;
  const long t2 = tiles_of(g, 128, splits, has_rows), t1 = tiles_of(g, 64, splits, has_rows);
  // With a live-row list the host does not know how many row tiles survive, and one workgroup per CU makes a partly
  // filled last round expensive (decode, 8921 live of 14592 rows x 1024: 280 workgroups = 1.1 rounds, 0.111 vs 0.081 ms
  // on the 128x128 kernel; x 1536: 0.148 vs 0.109): only launches of >= 4 estimated rounds take the wide kernel then.
  if (has_rows) return t2 >= 1024 ? 2 : 0;
  if (t2 >= 160) return 2;
  if (t1 >= 160 && kps >= 1024 && splits == 1) return 1;       // (with reduce splits the 64-row tile lost: 0.068 vs 0.061 ms)
  return 0;
}

template <class Epi>
static bool launch_wide_any(const GemmShape& g, const Epi& epi, int splits, bool has_rows, bool heavy, std::vector<std::pair<GemmShape, Epi>>* queue,
                            const char* name, hipStream_t s) {
  const int mt = wide_pick(g, splits, has_rows, 3, heavy);
  if (!mt) return false;
  if (queue && splits == 1 && g_wide_group_on) { queue->push_back({g, epi}); bf16_note_wide("gemm_wide_queued"); return true; }
  if (mt == 2) launch_wide_one<2, Epi>(g, epi, splits, s); else launch_wide_one<1, Epi>(g, epi, splits, s);
  note(mt, name, false);
  return true;
}
bool launch_wide_linear(const GemmShape& g, const EpiLinear& epi, bool has_rows, bool heavy, hipStream_t s) {
  return launch_wide_any<EpiLinear>(g, epi, 1, has_rows, heavy, &g_wide_lin, "EpiLinear", s);
}
bool launch_wide_dx(const GemmShape& g, const EpiStore& epi, int splits, bool has_rows, hipStream_t s) {
  return launch_wide_any<EpiStore>(g, epi, splits, has_rows, false, &g_wide_sto, "EpiStore", s);
}

extern "C" {
// Between begin and flush, forward / dX GEMMs that take the wide kernel are QUEUED; the flush launches the queued problems
// of each kind as one grid.  The caller guarantees they are independent of each other and of everything it launches in
// between (the two modalities of the visual encoder: models/Encoder.py:47-59 runs them one after the other).
int nacf_wide_group_begin(void) {
  g_wide_group_on = true;
  return NACF_OK;
}
int nacf_wide_group_flush(nacf_stream_t stream) {
  g_wide_group_on = false;
  int n = flush_one(g_wide_lin, "EpiLinear", as_hip(stream));
  n += flush_one(g_wide_sto, "EpiStore", as_hip(stream));
  NACF_LAUNCH_CHECK("nacf_wide_group_flush");
  return n;      // >= 0: number of GEMMs launched by the flush
}
}

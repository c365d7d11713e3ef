// Weight gradients on the 256 x 256 eight-phase body for fp32 operands (gemm_g256w.hpp) and their host launcher.
//
// Reference: the autograd of nn.Linear (models/bert.py:182-247, models/Encoder.py:9-25,62-66, models/__init__.py:83):
// dW[N][K] (+)= dZ[rows][N]^T X[rows][K] over the LIVE rows, db[N] (+)= column sums of dZ.
// g256_dw_group_kernel<NS>: every problem of a backward pass in ONE grid, one 256 x 256 output tile x one k-split per workgroup;
// the k-tiles (32 live rows each) of a problem are dealt round-robin to its splits (the live-row count is only known on the
// device: every split gets the same share whatever it is), longest walks first; a problem with one split writes dW and db
// itself (beta), the others leave fp32 slabs / bias partials for the caller's combine launch.  NS = 1: the throughput mode
// (operands rounded to bf16 in registers), NS = 3: the exact mode (three-term split, six products).
#include "gemm_g256_launch.hpp"
#include "gemm_g256w.hpp"
#include <algorithm>
#include <stdlib.h>

namespace {

#ifndef G256W_PHASES
#define G256W_PHASES 2      // phases per k-tile of the body (gemm_g256w.hpp: 4 | 2; 2 measured 6-11 % faster in both modes)
#endif
constexpr int G256_MAX_PROBS = 24;
constexpr int G256_MAX_SPLITS = 8;

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct DwProb {
  const float* A; const float* B; float* C;            // A = dZ [rows][lda], B = X [rows][ldb]
  float* bias;                                         // S == 1: db (beta) | S > 1: partials [S][I] | nullptr
  int64_t lda, ldb, ldc, slab_stride;                  // C: dW (S == 1) or slab 0 (S > 1; slab z at C + z * slab_stride, ldc = J)
  const int* list; const int* count;
  int M, I, J, tiles_j, tiles, S, wg0, vec;
  float beta;
};
struct DwTable { int n; int wg_end; int xcd_run; DwProb p[G256_MAX_PROBS]; };

struct EpiAcc {
  float* C; int64_t ldc; int rows, cols, vec; float beta;
  __device__ __forceinline__ void operator()(int r, int c, g256::f32x4 v) const {
    if (r >= rows || c >= cols) return;
    float* cp = C + (int64_t)r * ldc + c;
    if (vec && c + 3 < cols) {
      if (beta != 0.f) v += beta * *reinterpret_cast<const g256::f32x4*>(cp);
      *reinterpret_cast<g256::f32x4*>(cp) = v;
    } else {
      for (int e = 0; e < 4; ++e)
        if (c + e < cols) cp[e] = (beta != 0.f) ? v[e] + beta * cp[e] : v[e];
    }
  }
};
struct EpiBias {
  float* b; int rows; float beta;
  __device__ __forceinline__ void operator()(int r, float s) const {
    if (r < rows) b[r] = (beta != 0.f) ? s + beta * b[r] : s;
  }
};

// wave-uniform values the compiler cannot prove uniform (a table entry picked by a searched index): without this every buffer
// operation is wrapped in a waterfall loop
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* uni(T* q) {
  const uint64_t v = reinterpret_cast<uint64_t>(q);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t uni(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

template <int NS>
__global__ __launch_bounds__(512, 2) void g256_dw_group_kernel(DwTable t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Workgroups go to the 8 XCDs round-robin; each XCD has its own L2.  Inside windows of 8 * xcd_run workgroups the index is
  // permuted so that one XCD takes xcd_run CONSECUTIVE entries of the schedule (tiles of one problem and one split: they share
  // their operand panels), the windows keep the longest-first order of the schedule across the chip.
  int bid = (int)blockIdx.x;
  if (t.xcd_run > 1) {
    const int win = 8 * t.xcd_run, w0 = bid / win * win;
    if (w0 + win <= t.wg_end) {
      const int r = bid - w0;
      bid = w0 + (r & 7) * t.xcd_run + (r >> 3);
    }
  }
  int e = 0;
#pragma unroll 1
  while (e + 1 < t.n && bid >= t.p[e + 1].wg0) ++e;
  DwProb p;
  {
    const DwProb& q = t.p[e];
    p.A = uni(q.A); p.B = uni(q.B); p.C = uni(q.C); p.bias = uni(q.bias); p.lda = uni(q.lda); p.ldb = uni(q.ldb); p.ldc = uni(q.ldc);
    p.slab_stride = uni(q.slab_stride); p.list = uni(q.list); p.count = uni(q.count); p.M = uni(q.M); p.I = uni(q.I); p.J = uni(q.J);
    p.tiles_j = uni(q.tiles_j); p.tiles = uni(q.tiles); p.S = uni(q.S); p.wg0 = uni(q.wg0); p.vec = uni(q.vec); p.beta = uni(q.beta);
  }
  const int local = bid - p.wg0;
  const int z = local / p.tiles, tile = local - z * p.tiles;
  const int ti = tile / p.tiles_j, tj = tile - ti * p.tiles_j;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  g256w::Walk w;
  w.n_live = uni(p.count ? min(p.M, *p.count) : p.M);
  w.list = g256w::row_list(p.list, p.M); w.tile0 = z; w.step = p.S;
  const int nk_all = (w.n_live + g256w::BK - 1) / g256w::BK;
  w.nk = nk_all > z ? (nk_all - z + p.S - 1) / p.S : 0;
  if (w.nk < 1) w.nk = 1;      // an empty split still writes its (zero) tile: rows past the live count are fetched as zeros
  const int i0 = ti * g256w::BM, j0 = tj * g256w::BN;
  const g256w::Operand oa = g256w::operand(p.A, p.lda, i0, p.M, lane, wave), ob = g256w::operand(p.B, p.ldb, j0, p.M, lane, wave);
  EpiAcc epi;
  epi.C = p.C + (int64_t)z * p.slab_stride + (int64_t)i0 * p.ldc + j0;
  epi.ldc = p.ldc; epi.rows = p.I - i0; epi.cols = p.J - j0; epi.vec = p.vec; epi.beta = p.S == 1 ? p.beta : 0.f;
  EpiBias eb;
  eb.b = p.bias ? p.bias + (p.S == 1 ? 0 : (int64_t)z * p.I) + i0 : nullptr; eb.rows = p.I - i0; eb.beta = p.S == 1 ? p.beta : 0.f;
  if (p.bias && tj == 0) g256w::body<NS, true, G256W_PHASES>(smem, oa, ob, w, epi, eb);
  else g256w::body<NS, false, G256W_PHASES>(smem, oa, ob, w, epi, eb);
}

}  // namespace

// Which arithmetic modes take this path: both bf16 matrix-core modes.  Measured on the NACF step (128 videos, one box, interleaved
// runs): throughput mode 1.937 -> 1.732 ms per step, exact mode 2.645 -> 2.58 ms (the grouped launch 0.643 -> 0.58 ms = 0.46 of
// the mode's roof; with four phases per k-tile it tied at 2.652).  NACF_DW_G256 (A/B switch, read per call: a handful of calls per
// step, none inside a replayed graph): 0 = the 128 x 128 grouped kernel of gemm_bf16.hpp, 1 = throughput mode only, 3 = both (default).
bool g256_dw_enabled(int mode) {
  const char* e = getenv("NACF_DW_G256");
  const int v = e ? atoi(e) : 3;
  return mode == NACF_GEMM_BF16 ? v != 0 : (mode == NACF_GEMM_BF16X3 && v == 3);
}

static int g256_max_splits(int M, int N, int K, int ns) {
  const int tiles = cdiv(N, g256w::BM) * cdiv(K, g256w::BN), nk = cdiv(M, g256w::BK);
  const int min_walk = ns == 3 ? 6 : 16;      // k-tiles per split at least: a split costs a prologue and a slab
  int s = 256 / tiles;
  if (s > nk / min_walk) s = nk / min_walk;
  if (s > G256_MAX_SPLITS) s = G256_MAX_SPLITS;
  return s < 1 ? 1 : s;
}

size_t g256_dw_workspace(int M, int N, int K) {
  const int a = g256_max_splits(M, N, K, 1), b = g256_max_splits(M, N, K, 3), s = a > b ? a : b;
  return (s > 1 ? up256((size_t)s * N * K * 4) + up256((size_t)s * N * 4) : 0) + 256;
}

template <int NS>
static void launch_dw(const DwTable& dt, hipStream_t s) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(g256_dw_group_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, g256w::LDS_BYTES);
    raised = true;
  }
  hipLaunchKernelGGL(g256_dw_group_kernel<NS>, dim3(dt.wg_end), dim3(g256w::THREADS), g256w::LDS_BYTES, s, dt);
}

int g256_dw_group_launch(const G256DwItem* items, int n, int ns, std::vector<G256Reduce>& reduces, int* n_workgroups, hipStream_t s) {
  *n_workgroups = 0;
  for (int done = 0; done < n;) {
    const int m = std::min(n - done, G256_MAX_PROBS);
    const G256DwItem* it = items + done;
    // ---- the walk length: S_i = round(k-tiles_i / T) within its cap.  T = 32 k-tiles (1024 live rows per workgroup) is the
    // measured optimum of the NACF step's set in both modes (tools/dw_g256_bench.py with NACF_DW_G256_WALK = 12 .. 128:
    // throughput mode 0.428 / 0.409 / 0.400 / 0.412 / 0.423 / 0.523 ms at 12 / 24 / 32 / 48 / 64 / 128, exact mode 0.801 /
    // 0.781 / 0.786 / 0.840 / 0.880 / 1.016): ~2 rounds of workgroups, which the dispatcher balances, at 8 % fixed cost.
    // (Two cost models that pick T per group -- rounds x (longest walk + fixed), and the list-scheduling bound -- landed on
    // longer walks and lost 3-12 %.)
    std::vector<int> nk(m), tiles(m), cap(m), S(m);
    for (int i = 0; i < m; ++i) {
      const int m_eff = it[i].rows ? (int)((long)it[i].M * 29 / 50) : it[i].M;      // ~58 % of the slots are live (not known to the host)
      nk[i] = cdiv(m_eff > 0 ? m_eff : 1, g256w::BK);
      tiles[i] = cdiv(it[i].N, g256w::BM) * cdiv(it[i].K, g256w::BN);
      cap[i] = g256_max_splits(it[i].M, it[i].N, it[i].K, ns);
    }
    const char* xe = getenv("NACF_DW_G256_XCD_RUN");      // consecutive schedule entries per XCD (0 / 1: plain round-robin)
    // step set, graph replay, one box: throughput mode 0.372 / 0.348 / 0.351 / 0.362 / 0.420 ms at 0 / 4 / 8 / 16 / 32; exact mode 0.703 /
    // 0.702 / 0.692 / 0.782 / 0.825 (long runs unbalance the XCDs: the walks differ per problem)
    const int xcd_run = xe ? atoi(xe) : 4;
    const char* te = getenv("NACF_DW_G256_WALK");
    const int best_t = (te && atoi(te) > 0) ? atoi(te) : 32;
    // ... and no more slabs than `slab_mb` MB per problem: a split of a big output (the vocabulary projection: 21.6 MB) costs more
    // HBM traffic (slab write + combine read) than its shorter walk buys -- its long walks go first in the grid instead
    const char* se = getenv("NACF_DW_G256_SLAB_MB");
    const double slab_mb = (se && atof(se) > 0) ? atof(se) : 16.0;
    for (int i = 0; i < m; ++i) {
      const int by_bytes = std::max(1, (int)(slab_mb * 1048576.0 / ((double)it[i].N * it[i].K * 4.0)));
      S[i] = std::max(1, std::min(std::min((nk[i] + best_t / 2) / best_t, cap[i]), by_bytes));
    }
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
      const long wa = (long)nk[a] * S[b], wb = (long)nk[b] * S[a];      // k-tiles / splits, longest first
      return wa != wb ? wa > wb : a < b;
    });
    DwTable dt = {};
    int wg = 0;
    for (int oi = 0; oi < m; ++oi) {
      const int i = order[oi];
      const G256DwItem& x = it[i];
      NACF_CHECK(x.ws && x.ws_bytes >= g256_dw_workspace(x.M, x.N, x.K), NACF_EWORKSPACE, "nacf_linear_bwd_weight(g256): workspace too small");
      float* slabs = reinterpret_cast<float*>(x.ws);
      float* part = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(x.ws) + up256((size_t)S[i] * x.N * x.K * 4));
      DwProb& p = dt.p[dt.n++];
      p.A = x.dZ; p.B = x.X; p.lda = x.lddz; p.ldb = x.ldx; p.list = x.rows; p.count = x.count; p.M = x.M; p.I = x.N; p.J = x.K;
      p.tiles_j = cdiv(x.K, g256w::BN); p.tiles = tiles[i]; p.S = S[i]; p.wg0 = wg; p.beta = x.beta;
      if (S[i] == 1) {
        p.C = x.dW; p.ldc = x.lddw; p.slab_stride = 0; p.bias = x.db;
        p.vec = ((x.lddw & 3) == 0 && (reinterpret_cast<uintptr_t>(x.dW) & 15) == 0) ? 1 : 0;
      } else {
        p.C = slabs; p.ldc = x.K; p.slab_stride = (int64_t)x.N * x.K; p.vec = (x.K & 3) == 0 ? 1 : 0; p.bias = x.db ? part : nullptr;
        G256Reduce r;
        r.slabs = slabs; r.dW = x.dW; r.lddw = x.lddw; r.part = x.db ? part : nullptr; r.db = x.db; r.N = x.N; r.K = x.K;
        r.splits = S[i]; r.part_rows = S[i]; r.beta = x.beta;
        reduces.push_back(r);
      }
      wg += tiles[i] * S[i];
    }
    dt.wg_end = wg;
    dt.xcd_run = xcd_run;
    if (ns == 3) launch_dw<3>(dt, s);
    else launch_dw<1>(dt, s);
    NACF_LAUNCH_CHECK("nacf_dw_group_flush(g256 grouped gemm)");
    *n_workgroups += wg;
    done += m;
  }
  return NACF_OK;
}

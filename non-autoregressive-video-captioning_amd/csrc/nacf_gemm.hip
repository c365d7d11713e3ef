// C entry points built on the fp32 MFMA GEMM: nn.Linear forward/backward and
// the fused vocabulary projection + softmax-max used by NA decoding.
#include "gemm_f32.hpp"
#include "gemm_bf16_launch.hpp"
#include "gemm_g256_launch.hpp"
#include <stdlib.h>
#include <mutex>
#include <algorithm>
#include <string.h>
#include <vector>

namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- arithmetic mode of every GEMM entry point (process-wide; include/nacf_hip.h NACF_GEMM_*)
int g_mode = NACF_GEMM_DEFAULT_MODE;
int gemm_mode() {
  const char* e = getenv("NACF_GEMM_MODE");   // tuning / test knob, read per call: 0 | 1 | 3 (or f32 | bf16 | bf16x3)
  if (e && *e) {
    if (!strcmp(e, "f32") || !strcmp(e, "0")) return NACF_GEMM_F32;
    if (!strcmp(e, "bf16") || !strcmp(e, "1")) return NACF_GEMM_BF16;
    if (!strcmp(e, "bf16x3") || !strcmp(e, "3")) return NACF_GEMM_BF16X3;
  }
  return g_mode;
}
inline int kalign() { return gemm_mode() == NACF_GEMM_F32 ? 16 : 32; }   // k-tile depth of the kernel family in use

thread_local char g_last_f32_kernel[160] = "";
thread_local bool g_last_was_bf16 = false;

// ---- registry of pre-split bf16 weight images (nacf_wimage_register): one entry per weight matrix
struct ImgMat { const float* w; int N, K; int64_t ld; const unsigned short* img; int64_t plane;
                const unsigned short* imgT; int64_t planeT; int ns; };
std::vector<ImgMat> g_imgs;

// forward image of W [N, K] (row pitch ldw): W is a registered matrix or a run of whole rows of one (packed q|k|v
// weights used slice-wise).  k-tile-major: row n of k-tile t at img[(t * N_registered + n) * 32]
bool find_image(const float* W, int64_t ldw, int N, int K, int ns, GemmShape& g) {
  for (const ImgMat& m : g_imgs) {
    if (m.ns != ns || !m.img || m.ld != ldw || m.K != K || W < m.w) continue;
    const int64_t off = W - m.w;
    if (off % m.ld != 0) continue;
    const int64_t n0 = off / m.ld;
    if (n0 + N > m.N) continue;
    g.Pimg = m.img + n0 * 32; g.ldpi = (int64_t)m.N * 32; g.pimg_plane = m.plane;
    return true;
  }
  return false;
}
// transposed image of exactly this matrix (dX = dZ W: P = W^T, rows = the K output columns, reduce over N)
bool find_image_t(const float* W, int64_t ldw, int N, int K, int ns, GemmShape& g) {
  for (const ImgMat& m : g_imgs) {
    if (m.w == W && m.N == N && m.K == K && m.ld == ldw && m.ns == ns && m.imgT) {
      g.Pimg = m.imgT; g.ldpi = (int64_t)m.K * 32; g.pimg_plane = m.planeT;
      return true;
    }
  }
  return false;
}

// tile selection: 0 = 128x128, 1 = 64x64.  Override with NACF_GEMM_TILE=128|64.
int forced_tile() {
  const char* e = getenv("NACF_GEMM_TILE");  // tuning / test knob, read per call
  return e ? (atoi(e) == 64 ? 1 : (atoi(e) == 128 ? 0 : -1)) : -1;
}

int pick_tile(int M, int N, int splits, bool has_rows = false, bool heavy_epilogue = false) {
  const int forced = forced_tile();
  if (forced >= 0) return forced;
  const long big = (long)cdiv(M, 128) * cdiv(N, 128) * splits;
  // a transcendental activation in the epilogue (gelu: one tanh per output) keeps the VALU busy for a long time per
  // tile; with only 2 big-tile workgroups per CU the matrix pipe idles meanwhile, 5-6 small-tile workgroups overlap it
  // (decode FFN 14592x2048x512 with gelu_new: 69.7 TF on 128x128, 94.9 TF on 64x64)
  if (heavy_epilogue) return big >= 8192 ? 0 : 1;
  // >= 3 workgroups per CU with the big tile, else go small (tools/gemm_bench.py: 64x64 wins or ties up to
  // 640 big tiles, 128x128 wins from 960).  With a live-row list the host does not know how many row tiles
  // survive (typically 40-60 %), so ask for twice the tiles.
  return big >= (has_rows ? 1536 : 768) ? 0 : 1;
}

// The bf16 matrix-core kernels (gemm_bf16.hpp) want other tiles than the fp32 ones: the 128x128 kernel amortises the
// staging work (convert / split + LDS stores + load issue, the longer phase of a k-tile) over 4x the MFMAs but runs
// only two workgroups per CU.  Measured (tools/gemm_bench.py --modes, tools/dw_rows_bench.py; profiles/r02_*):
// with >= ~200 big tiles the big tile wins (forward: 136 vs 112 TF exact at 240 tiles, K = 2048), a live-row
// forward needs ~400 (the row list costs the big tile more edge work), dX in the throughput mode from ~90.
// kind: 0 forward, 1 dX.  ns: 1 throughput, 3 exact.
int pick_tile_bf16(int kind, int M, int N, int splits, bool has_rows, int ns, bool heavy_epilogue = false) {
  const int forced = forced_tile();
  if (forced >= 0) return forced;
  const int m_eff = has_rows ? (int)((long)M * 29 / 50) : M;       // ~58 % of the slots are live (not known to the host)
  const long big = (long)cdiv(m_eff > 0 ? m_eff : 1, 128) * cdiv(N, 128) * splits;
  long thr;
  if (kind == 0) {
    // throughput mode: the small-tile kernel runs 7-8 workgroups per CU and hides a row list's edge work and a
    // transcendental epilogue (tanh|sigmoid, gelu) far better: 64x64 unless the launch is huge (the vocabulary)
    thr = has_rows ? (ns == 1 ? 1500 : 400) : (ns == 1 ? 300 : 200);
    if (heavy_epilogue && ns == 1) thr = 1L << 40;
  } else {
    thr = (ns == 1) ? 300 : (has_rows ? 300 : 200);      // tools/gemm_bench.py --modes bf16x3,bf16 --tiles 64,128,auto --images
  }
  return big >= thr ? 0 : 1;
}

template <class Epi> const char* epi_name();
template <> const char* epi_name<EpiLinear>() { return "EpiLinear"; }
template <> const char* epi_name<EpiStore>() { return "EpiStore"; }
template <> const char* epi_name<EpiArgmax>() { return "EpiArgmax"; }

template <bool QKC, bool PKC, class Epi>
void launch_gemm(const GemmShape& g0, const Epi& epi, int splits, int tile, bool vec, hipStream_t s) {
  GemmShape g = g0;
  snprintf(g_last_f32_kernel, sizeof(g_last_f32_kernel), "gemm_f32_kernel<%d, %d, 2, 2, %s, %s, %s, %s>", tile == 0 ? 128 : 64,
           tile == 0 ? 128 : 64, QKC ? "true" : "false", PKC ? "true" : "false", vec ? "true" : "false", epi_name<Epi>());
  g_last_was_bf16 = false;
  if (tile == 0) {
    g.tiles_m = cdiv(g.M, 128);
    g.tiles_n = cdiv(g.N, 128);
    // live row tiles + dead row tiles can be one more than the dense count (both partly filled)
    dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2, QKC, PKC, true, Epi>), grid, dim3(256), 0, s, g, epi);
    else hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2, QKC, PKC, false, Epi>), grid, dim3(256), 0, s, g, epi);
  } else {
    g.tiles_m = cdiv(g.M, 64);
    g.tiles_n = cdiv(g.N, 64);
    dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 2, 2, QKC, PKC, true, Epi>), grid, dim3(256), 0, s, g, epi);
    else hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 2, 2, QKC, PKC, false, Epi>), grid, dim3(256), 0, s, g, epi);
  }
}

inline void set_rows(GemmShape& g, const nacf_rowset* rs) {
  g.rows = rs ? rs->rows : nullptr;
  g.count = rs ? rs->count : nullptr;
  g.zero_dead = (rs && rs->zero_dead) ? 1 : 0;
}

// dst[rows[r]][c] = beta*dst + sum_z slab[z][rows[r]][c] over the live rows only (split-K combine of dX);
// zero_dead: the rows listed after the live ones get zeros (beta == 0) in the same pass
__global__ void splitk_reduce_rows_kernel(const float* __restrict__ slabs, int64_t slab_stride, int splits,
                                          float* __restrict__ dst, int64_t ldd, int rows, int cols, float beta,
                                          const int* __restrict__ live, const int* __restrict__ count, int zero_dead) {
  const int n_live = count ? min(rows, *count) : rows;
  const int n_do = (zero_dead && live && beta == 0.f) ? rows : n_live;
  const int64_t total = (int64_t)n_do * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx / cols), c = (int)(idx % cols);
    const int pr = live ? live[r] : r;
    float* d = dst + (int64_t)pr * ldd + c;
    if (r >= n_live) { *d = 0.f; continue; }
    float acc = 0.f;
    for (int zz = 0; zz < splits; ++zz) acc += slabs[(int64_t)zz * slab_stride + (int64_t)pr * cols + c];
    *d = (beta != 0.f) ? acc + beta * (*d) : acc;
  }
}

// dst[i] = beta*dst[i] + sum_z slab[z][i], fixed z order (deterministic split-K combine); optionally the same
// for the bias-gradient partials the dW GEMM left in part[z][n] (db[n] = beta*db[n] + sum_z part[z][n]).
// VEC: cols % 4 == 0 and 16-byte aligned rows -> float4 per thread.
template <bool VEC>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ slabs, int64_t slab_stride, int splits,
                                                   float* __restrict__ dst, int64_t ldd, int rows, int cols, float beta,
                                                   const float* __restrict__ part, float* __restrict__ db, int64_t gid,
                                                   int64_t gstride, int part_rows = -1) {
  if (part_rows < 0) part_rows = splits;      // the bias-gradient partials: one row per split unless told otherwise
  if (splits <= 0) {
    // (nothing to combine: the GEMM wrote dW itself; only the bias gradient is left)
  } else if constexpr (VEC) {
    const int c4n = cols >> 2;
    const int64_t total = (int64_t)rows * c4n;
    for (int64_t idx = gid; idx < total; idx += gstride) {
      const int r = (int)(idx / c4n), c = (int)(idx % c4n) * 4;
      const float* sp = slabs + (int64_t)r * cols + c;
      f32x4 acc = *reinterpret_cast<const f32x4*>(sp);
      for (int zz = 1; zz < splits; ++zz) acc += *reinterpret_cast<const f32x4*>(sp + (int64_t)zz * slab_stride);
      f32x4* d = reinterpret_cast<f32x4*>(dst + (int64_t)r * ldd + c);
      *d = (beta != 0.f) ? acc + beta * (*d) : acc;
    }
  } else {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t idx = gid; idx < total; idx += gstride) {
      const int r = (int)(idx / cols), c = (int)(idx % cols);
      float acc = 0.f;
      for (int zz = 0; zz < splits; ++zz) acc += slabs[(int64_t)zz * slab_stride + idx];
      float* d = dst + (int64_t)r * ldd + c;
      *d = (beta != 0.f) ? acc + beta * (*d) : acc;
    }
  }
  if (db) {
    for (int64_t n = gid; n < rows; n += gstride) {
      float acc = 0.f;
      for (int zz = 0; zz < part_rows; ++zz) acc += part[(int64_t)zz * rows + n];
      db[n] = (beta != 0.f) ? acc + beta * db[n] : acc;
    }
  }
}

template <bool VEC>
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int64_t slab_stride, int splits,
                                     float* __restrict__ dst, int64_t ldd, int rows, int cols, float beta,
                                     const float* __restrict__ part, float* __restrict__ db) {
  splitk_reduce_body<VEC>(slabs, slab_stride, splits, dst, ldd, rows, cols, beta, part, db,
                          (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// ---- deferred combines (nacf_dw_group_begin / _flush): the split-K combines of up to DW_GROUP_MAX dW GEMMs in ONE
// launch.  Every combine is a ~5-10 us kernel of its own otherwise (12 per NACF training step); their sums do not
// depend on each other, only on their own GEMM, so they can all wait for the end of the backward pass.  The
// arithmetic of each combine is exactly splitk_reduce_kernel's (same slab order), so results are bit-identical.
constexpr int DW_GROUP_MAX = 32;
struct DwReduceDesc {
  const float* slabs;
  float* dst;
  const float* part;
  float* db;
  int64_t slab_stride, ldd;
  int splits, rows, cols, vec;
  float beta;
  int block0;        // first block of this combine in the grouped launch
  int part_rows;     // rows of `part` (-1: one per split)
};
struct DwReduceTable {
  int n;
  int block_end;
  DwReduceDesc d[DW_GROUP_MAX];
};
__global__ void dw_group_reduce_kernel(DwReduceTable t) {
  int e = 0;
#pragma unroll 1
  while (e + 1 < t.n && (int)blockIdx.x >= t.d[e + 1].block0) ++e;
  const DwReduceDesc& d = t.d[e];
  const int nblk = (e + 1 < t.n ? t.d[e + 1].block0 : t.block_end) - d.block0;
  const int64_t gid = (int64_t)((int)blockIdx.x - d.block0) * blockDim.x + threadIdx.x;
  const int64_t gstride = (int64_t)nblk * blockDim.x;
  if (d.vec) splitk_reduce_body<true>(d.slabs, d.slab_stride, d.splits, d.dst, d.ldd, d.rows, d.cols, d.beta, d.part, d.db, gid, gstride, d.part_rows);
  else splitk_reduce_body<false>(d.slabs, d.slab_stride, d.splits, d.dst, d.ldd, d.rows, d.cols, d.beta, d.part, d.db, gid, gstride, d.part_rows);
}

// merge the per-tile (max, idx, sumexp) partials and apply the decode bookkeeping
constexpr int MERGE_NPL = 3;      // tile partials per lane held in registers: up to 192 column tiles (V <= 12288 at 64 columns a tile)
__global__ void argmax_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                    const int* __restrict__ pidx, int tiles_n, int rows,
                                    const int* __restrict__ live, const int* __restrict__ count,
                                    const int64_t* __restrict__ pad_tokens, int zero_mask_prob,
                                    const uint8_t* __restrict__ update_mask,
                                    int64_t* __restrict__ tokens, float* __restrict__ probs) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // logical row
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  if (count && row >= *count) return;
  // slot -> <pad> / update flags is a chain of dependent loads: start it here, under the merge (it used to follow it); a row whose
  // token is not to be replaced (mask-predict iterations: every position that is not re-masked) needs no merge at all
  const int prow = live ? live[row] : row;     // physical slot
  if (update_mask && !update_mask[prow]) return;
  const bool is_pad = pad_tokens && pad_tokens[prow] == NACF_PAD;
  float best = -3.0e38f;
  int bidx = 0x7fffffff;
  float s = 0.f;
  if (tiles_n <= 64 * MERGE_NPL) {
    // every partial of the row is requested before the first one is used (three chains of loads otherwise: maxima, then sums)
    float v[MERGE_NPL], e[MERGE_NPL];
    int ix[MERGE_NPL];
#pragma unroll
    for (int u = 0; u < MERGE_NPL; ++u) {
      const int t = lane + 64 * u;
      const int64_t at = (int64_t)(t < tiles_n ? t : tiles_n - 1) * rows + row;
      v[u] = pmax[at]; ix[u] = pidx[at]; e[u] = psum[at];
    }
#pragma unroll
    for (int u = 0; u < MERGE_NPL; ++u)
      if (lane + 64 * u < tiles_n && (v[u] > best || (v[u] == best && ix[u] < bidx))) { best = v[u]; bidx = ix[u]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bidx, o, 64);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
#pragma unroll
    for (int u = 0; u < MERGE_NPL; ++u)
      if (lane + 64 * u < tiles_n) s += e[u] * __expf(v[u] - best);
  } else {
    for (int t = lane; t < tiles_n; t += 64) {
      float v = pmax[(int64_t)t * rows + row];
      int i = pidx[(int64_t)t * rows + row];
      if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bidx, o, 64);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    for (int t = lane; t < tiles_n; t += 64)
      s += psum[(int64_t)t * rows + row] * __expf(pmax[(int64_t)t * rows + row] - best);
  }
  s = wave_sum(s);
  if (lane == 0) {
    int64_t tok = bidx;
    float p = 1.0f / s;
    if (is_pad) { tok = NACF_PAD; p = 1.0f; }
    if (zero_mask_prob && tok == NACF_MASK) p = 0.f;
    tokens[prow] = tok; probs[prow] = p;
  }
}

// training form of the merge: lse = log sum_n exp(z[n]) per live row from the per-tile (max, sum-exp) pairs, the arg-max
// (accuracy meter) and log p(label) = z[label] - lse read straight from the stored logits
__global__ void lse_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                 const int* __restrict__ pidx, int tiles_n, int rows, const int* __restrict__ live,
                                 const int* __restrict__ count, const float* __restrict__ logits, int64_t ldl,
                                 const int64_t* __restrict__ labels, float* __restrict__ lse,
                                 int64_t* __restrict__ argmax, float* __restrict__ label_logp) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // logical row
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  if (count && row >= *count) return;
  // slot -> label -> that label's logit is a chain of three dependent loads: start it here, under the merge (it used to follow it)
  const int prow = live ? live[row] : row;     // physical slot
  const int64_t lab = (label_logp && labels) ? labels[prow] : -1;
  const float lab_logit = lab >= 0 ? logits[(int64_t)prow * ldl + lab] : 0.f;
  float best = -3.0e38f;
  int bidx = 0x7fffffff;
  float s = 0.f;
  if (tiles_n <= 64 * MERGE_NPL) {
    // every partial of the row is requested before the first one is used (three chains of loads otherwise: maxima, then sums)
    float v[MERGE_NPL], e[MERGE_NPL];
    int ix[MERGE_NPL];
#pragma unroll
    for (int u = 0; u < MERGE_NPL; ++u) {
      const int t = lane + 64 * u;
      const int64_t at = (int64_t)(t < tiles_n ? t : tiles_n - 1) * rows + row;
      v[u] = pmax[at]; ix[u] = pidx[at]; e[u] = psum[at];
    }
#pragma unroll
    for (int u = 0; u < MERGE_NPL; ++u)
      if (lane + 64 * u < tiles_n && (v[u] > best || (v[u] == best && ix[u] < bidx))) { best = v[u]; bidx = ix[u]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bidx, o, 64);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
#pragma unroll
    for (int u = 0; u < MERGE_NPL; ++u)
      if (lane + 64 * u < tiles_n) s += e[u] * expf(v[u] - best);
  } else {
    for (int t = lane; t < tiles_n; t += 64) {
      float v = pmax[(int64_t)t * rows + row];
      int i = pidx[(int64_t)t * rows + row];
      if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bidx, o, 64);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    for (int t = lane; t < tiles_n; t += 64) s += psum[(int64_t)t * rows + row] * expf(pmax[(int64_t)t * rows + row] - best);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float l = best + logf(s);
    if (lse) lse[prow] = l;
    if (argmax) argmax[prow] = bidx;
    if (label_logp && labels) label_logp[prow] = (lab >= 0) ? lab_logit - l : 0.f;
  }
}

// partition: rows[0..count) = ascending i with (tokens ? tokens[i] != PAD) && (flags ? flags[i] != 0); the other
// slots fill rows[count..n) from the END (so they come out descending -- their order is irrelevant, the GEMMs only
// zero-fill them).  One workgroup, ONE sweep, 4 consecutive slots per thread, wave scans by shuffle.
// EPT consecutive slots per thread and sweep.  (16 -- the decode canvas' 15360 slots in one sweep instead of four -- is slower, 18 us
// against 11: a lane then reads 128 consecutive bytes and every load instruction touches 64 cache lines.  All sweeps' loads and wave
// scans up front with ONE barrier to place them: 12.6 us against 11.1, and 8.3 against 5.5 for the training step's single sweep.)
// The token / flag loads are unconditional (past the end: the last slot is read and not used) so that a thread's loads go out together.
template <int EPT>
__global__ __launch_bounds__(1024) void rowset_build_kernel(const int64_t* __restrict__ tokens,
                                                             const uint8_t* __restrict__ flags, int n,
                                                             int* __restrict__ rows, int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 1024 * EPT) {
    const int i0 = c0 + threadIdx.x * EPT;
    int64_t tok[EPT];
    uint8_t flg[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int ic = i0 + e < n ? i0 + e : n - 1;
      tok[e] = tokens ? tokens[ic] : 1;
      flg[e] = flags ? flags[ic] : 1;
    }
    unsigned live = 0u;
    int c = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const bool l = (i0 + e < n) && (!tokens || tok[e] != NACF_PAD) && (!flags || flg[e] != 0);
      live |= (l ? 1u : 0u) << e;
      c += l ? 1 : 0;
    }
    int incl = c;                       // inclusive scan of the per-thread live counts inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    int lpos = off + incl - c;          // live slots before this thread's first element
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = i0 + e;
      if (i < n) {
        if ((live >> e) & 1u) rows[lpos++] = i;
        else rows[n - 1 - (i - lpos)] = i;     // i - lpos = dead slots before i
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = base_s;
}

// A weight gradient too small for the matrix-core kernels: N < 128 output rows cannot join the grouped launch, and the 64x64
// kernel spent 17 us on the length head's second Linear (dW [20, 512] over 128 rows: 8 workgroups on its checked-load path,
// 2.3 TF -- VERDICT round 3).  One thread per output element, an fp32 fmaf chain over the (live) rows in order: exact fp32
// products, deterministic; the thread of column 0 also sums its row's bias gradient.
__global__ __launch_bounds__(256) void dw_small_kernel(const float* __restrict__ dZ, int64_t lddz, const float* __restrict__ X,
                                                       int64_t ldx, float* __restrict__ dW, int64_t lddw, float* __restrict__ db,
                                                       int M, int N, int K, float beta, const int* __restrict__ rows,
                                                       const int* __restrict__ count) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int n = idx / K, k = idx - n * K;
  const int Mlive = rows ? min(M, *count) : M;
  float acc = 0.f, bsum = 0.f;
  const bool do_b = db && k == 0;
#pragma unroll 32      // (the loads of a body go out together: two latencies -- row index, then operands -- per body, not per row)
  for (int i = 0; i < Mlive; ++i) {
    const int m = rows ? rows[i] : i;
    const float z = dZ[(int64_t)m * lddz + n];
    acc = fmaf(z, X[(int64_t)m * ldx + k], acc);
    bsum += z;
  }
  float* d = dW + (int64_t)n * lddw + k;
  *d = (beta != 0.f) ? acc + beta * (*d) : acc;
  if (do_b) db[n] = (beta != 0.f) ? bsum + beta * db[n] : bsum;
}

}  // namespace

extern "C" {

int nacf_rowset_build(const int64_t* tokens, const uint8_t* flags, int64_t n, int32_t* rows, int32_t* count,
                      nacf_stream_t stream) {
  NACF_CHECK((tokens || flags) && rows && count && n > 0 && n < 0x7fffffffLL, NACF_EINVAL, "nacf_rowset_build: bad argument");
  hipLaunchKernelGGL(rowset_build_kernel<4>, dim3(1), dim3(1024), 0, as_hip(stream), tokens, flags, (int)n, rows, count);
  NACF_LAUNCH_CHECK("nacf_rowset_build");
  return NACF_OK;
}

int nacf_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, float* Y, int64_t ldy,
                    int M, int N, int K, const nacf_epilogue* ep, const nacf_rowset* rs, nacf_stream_t stream) {
  NACF_CHECK(X && W && Y, NACF_EINVAL, "nacf_linear_fwd: null pointer");
  NACF_CHECK(M > 0 && N > 0 && K > 0, NACF_EINVAL, "nacf_linear_fwd: bad shape M=%d N=%d K=%d", M, N, K);
  NACF_CHECK(ldx >= K && ldw >= K && ldy >= N, NACF_EINVAL, "nacf_linear_fwd: leading dimension too small");
  NACF_CHECK(!rs || (rs->rows && rs->count), NACF_EINVAL, "nacf_linear_fwd: incomplete row set");
  EpiLinear epi;
  memset(&epi, 0, sizeof(epi));
  epi.Y = Y;
  epi.ldy = ldy;
  if (ep) epi.ep = *ep;
  NACF_CHECK(!((epi.ep.p_drop1 > 0.f || epi.ep.p_drop2 > 0.f) && !epi.ep.rng_state), NACF_EINVAL,
             "nacf_linear_fwd: dropout requested without rng_state");
  NACF_CHECK(epi.ep.p_drop1 < 1.f && epi.ep.p_drop2 < 1.f, NACF_EINVAL, "nacf_linear_fwd: p_drop must be < 1");
  bool vo = (ldy % 4 == 0) && aligned16(Y);
  if (epi.ep.preact) vo = vo && (epi.ep.ld_preact % 4 == 0) && aligned16(epi.ep.preact);
  if (epi.ep.residual) vo = vo && (epi.ep.ld_residual % 4 == 0) && aligned16(epi.ep.residual);
  epi.vec_out = vo ? 1 : 0;
  // the interior-tile epilogue wants a float4-addressable bias and (for dropout) 4-aligned Philox groups
  const bool no_drop = !(epi.ep.p_drop1 > 0.f) && !(epi.ep.p_drop2 > 0.f);
  epi.vec_bias = ((!epi.ep.bias || aligned16(epi.ep.bias)) && (no_drop || N % 4 == 0)) ? 1 : 0;
  GemmShape g = {};
  g.Q = X; g.P = W; g.ldq = ldx; g.ldp = ldw; g.M = M; g.N = N; g.K = K;
  g.k_per_split = cdiv(K, 32) * 32;
  set_rows(g, rs);
  const bool vec = (ldx % 4 == 0) && (ldw % 4 == 0) && aligned16(X) && aligned16(W);
  const bool heavy = epi.ep.act == NACF_ACT_GELU_NEW || epi.ep.act == NACF_ACT_GELU_ERF || epi.ep.act == NACF_ACT_TANH ||
                     epi.ep.act == NACF_ACT_SIGMOID || epi.ep.act == NACF_ACT_TANH_SIGMOID;
  const int mode = gemm_mode();
  if (mode != NACF_GEMM_F32 && vec) {          // bf16 matrix cores (16-byte addressable operands only)
    find_image(W, ldw, N, K, mode, g);
    const bool heavy_w = heavy || epi.ep.p_drop1 > 0.f || epi.ep.p_drop2 > 0.f;      // nothing overlaps the wide kernel's epilogue
    if (!(mode == NACF_GEMM_BF16X3 && launch_wide_linear(g, epi, rs != nullptr, heavy_w, as_hip(stream))) &&
        !(mode == NACF_GEMM_BF16X3 && launch_dma128_linear(g, epi, rs != nullptr, as_hip(stream))))
      launch_bf16_linear(g, epi, pick_tile_bf16(0, M, N, 1, rs != nullptr, mode, heavy), mode, as_hip(stream));
    g_last_was_bf16 = true;
  } else {
    launch_gemm<true, true, EpiLinear>(g, epi, 1, pick_tile(M, N, 1, rs != nullptr, heavy), vec, as_hip(stream));
  }
  NACF_LAUNCH_CHECK("nacf_linear_fwd");
  return NACF_OK;
}

// dX = dZ W with a long reduce dimension and a narrow output (the vocabulary projection: N = V,
// K = D) has too few output tiles to fill 256 CUs: split the reduce dimension over workgroups.
static int bwd_data_splits(int M, int N, int K, bool has_rows) {
  { const char* e = getenv("NACF_GEMM_SPLITS"); if (e && atoi(e) > 0) return atoi(e); }
  // output tiles the 64x64 kernel would have (a live-row list typically keeps ~55 % of the row tiles)
  long tiles = (long)cdiv(M, 64) * cdiv(K, 64);
  if (has_rows) tiles = tiles * 11 / 20;
  if (N >= 4096) {   // the vocabulary projection: long reduce dimension, narrow output
    if (tiles >= 1024) return 1;
    int s = (int)((1536 + tiles - 1) / tiles);
    const int max_s = N / 1024;
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
  }
  // fewer than two output tiles per CU and a reduce dimension worth halving (tools/sweep_dx_splits.sh:
  // 2980x512 <- 2048: 92 -> 77 us, <- 1536: 70 -> 61 us; <- 512 loses)
  return (tiles < 512 && N >= 1024) ? 2 : 1;
}

size_t nacf_linear_bwd_data_workspace(int M, int N, int K) {
  const int a = bwd_data_splits(M, N, K, false), b = bwd_data_splits(M, N, K, true);
  const int s = a > b ? a : b;
  return (s > 1 ? (size_t)s * M * K * sizeof(float) : 0) + 256;
}

int nacf_linear_bwd_data(const float* dZ, int64_t lddz, const float* W, int64_t ldw, float* dX, int64_t lddx,
                         int M, int N, int K, float beta, void* ws, size_t ws_bytes, const nacf_rowset* rs,
                         nacf_stream_t stream) {
  NACF_CHECK(dZ && W && dX, NACF_EINVAL, "nacf_linear_bwd_data: null pointer");
  NACF_CHECK(M > 0 && N > 0 && K > 0, NACF_EINVAL, "nacf_linear_bwd_data: bad shape");
  NACF_CHECK(lddz >= N && ldw >= K && lddx >= K, NACF_EINVAL, "nacf_linear_bwd_data: leading dimension too small");
  NACF_CHECK(!rs || (rs->rows && rs->count), NACF_EINVAL, "nacf_linear_bwd_data: incomplete row set");
  // dX[m][k] = sum_n dZ[m][n] W[n][k]: Q = dZ (KC, reduce = n), P rows = k, P element (k, n) at W[n*ldw + k] (MC)
  const int splits = bwd_data_splits(M, N, K, rs != nullptr);
  NACF_CHECK(splits == 1 || (ws && ws_bytes >= nacf_linear_bwd_data_workspace(M, N, K) && aligned16(ws)), NACF_EWORKSPACE,
             "nacf_linear_bwd_data: workspace too small / misaligned");
  hipStream_t s = as_hip(stream);
  GemmShape g = {};
  g.Q = dZ; g.P = W; g.ldq = lddz; g.ldp = ldw; g.M = M; g.N = K; g.K = N;
  const bool vec = (lddz % 4 == 0) && (ldw % 4 == 0) && aligned16(dZ) && aligned16(W);
  const int mode = gemm_mode();
  const bool bf16 = mode != NACF_GEMM_F32 && vec;
  const int ka = bf16 ? 32 : 16;
  g.k_per_split = cdiv(cdiv(N, splits), ka) * ka;
  set_rows(g, rs);
  const int real_splits = cdiv(N, g.k_per_split);
  EpiStore epi;
  if (real_splits > 1) {
    epi.C = reinterpret_cast<float*>(ws); epi.ldc = K; epi.beta = 0.f; epi.slab_stride = (int64_t)M * K;
    epi.vec_out = (K % 4 == 0) ? 1 : 0;
    g.zero_dead = 0;   // the combine kernel zeroes the dead rows of dX, not the GEMM those of the slabs
  } else {
    epi.C = dX; epi.ldc = lddx; epi.beta = beta; epi.slab_stride = 0;
    epi.vec_out = ((lddx % 4 == 0) && aligned16(dX)) ? 1 : 0;
  }
  if (bf16) {
    find_image_t(W, ldw, N, K, mode, g);       // P = W^T image [K, N] when registered, else the fp32 W read transposed
    if (!(mode == NACF_GEMM_BF16X3 && launch_wide_dx(g, epi, real_splits, rs != nullptr, s)) &&
        !(mode == NACF_GEMM_BF16X3 && launch_dma128_dx(g, epi, real_splits, rs != nullptr, s)))
      launch_bf16_dx(g, epi, real_splits, pick_tile_bf16(1, M, K, real_splits, rs != nullptr, mode), mode, s);
    g_last_was_bf16 = true;
  } else {
    launch_gemm<true, false, EpiStore>(g, epi, real_splits, pick_tile(M, K, real_splits, rs != nullptr), vec, s);
  }
  NACF_LAUNCH_CHECK("nacf_linear_bwd_data");
  if (real_splits > 1) {
    const int64_t total = (int64_t)M * K;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_rows_kernel, dim3(blocks), dim3(256), 0, s, epi.C, (int64_t)M * K, real_splits, dX,
                       lddx, M, K, beta, rs ? rs->rows : nullptr, rs ? rs->count : nullptr, (rs && rs->zero_dead) ? 1 : 0);
    NACF_LAUNCH_CHECK("nacf_linear_bwd_data(reduce)");
  }
  return NACF_OK;
}

// dW on the bf16 matrix cores: the 128x128 tile from 16 output tiles on (132 vs 90 TF exact, 134 vs 91 bf16 on the
// FFN weights with 2980 live rows); splits: the fewest that give >= 480 workgroups with a well-filled last round
// (512 resident workgroups per round), each split walking >= 160 reduce rows (5 k-tiles).
static int bwd_weight_splits_bf16(int M, int N, int K, bool has_rows, int* tile_out) {
  const int m_eff = has_rows ? M * 11 / 20 : M;
  int tile = ((long)cdiv(N, 128) * cdiv(K, 128) >= 16 && m_eff >= 1024) ? 0 : 1;
  const int forced = forced_tile();
  if (forced >= 0) tile = forced;
  const int t = tile == 0 ? 128 : 64;
  const long tiles = (long)cdiv(N, t) * cdiv(K, t);
  const long slots = tile == 0 ? 512 : 768;
  int max_s = m_eff / 160 > 0 ? m_eff / 160 : 1;
  if (max_s > 64) max_s = 64;
  int s = 1;
  for (; s < max_s; ++s) {
    const long wgs = tiles * s;
    const long rounds = (wgs + slots - 1) / slots;
    if (wgs >= slots * 15 / 16 && wgs * 100 >= rounds * slots * 85) break;
  }
  { const char* e = getenv("NACF_GEMM_SPLITS"); if (e && atoi(e) > 0) s = atoi(e); }
  *tile_out = tile;
  return s;
}

static int bwd_weight_splits(int M, int N, int K, bool has_rows, int* tile_out) {
  // dW = dZ^T X: small output (N x K), long reduce dimension (M rows, ~55 % of them live under a row list).
  // The reduce-dimension split supplies the parallelism.  tools/sweep_dw_splits.sh: the 128x128 tile wins
  // only with >= 32 of them AND a long reduce walk (7680+ rows); otherwise 64x64 with ~768 workgroups.
  const int m_eff = has_rows ? M * 11 / 20 : M;
  int tile = ((long)cdiv(N, 128) * cdiv(K, 128) >= 32 && m_eff >= 4096) ? 0 : 1;
  const int forced = forced_tile();
  if (forced >= 0) tile = forced;
  const int t = tile == 0 ? 128 : 64;
  const long tiles = (long)cdiv(N, t) * cdiv(K, t);
  const long want = tile == 0 ? 512 : 768;   // tools/dw_rows_bench.py: 192 tiles x 4, 64 x 12, 256 x 3 beat x 6 / x 16 / x 4
  int s = (int)((want + tiles - 1) / tiles);
  // just over one round of workgroups (1280 resident 64-tiles) with a long reduce walk: the few tiles of the second round
  // would run alone for a full walk -- halve the walks instead (vocabulary dW, 1320 tiles x 2311 live rows: 0.351 -> 0.315 ms)
  if (tile == 1 && tiles >= 1024 && tiles < 2048 && m_eff >= 2048) s = 2;
  const int max_s = m_eff / 128 > 0 ? m_eff / 128 : 1;   // >= 8 k-tiles per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  { const char* e = getenv("NACF_GEMM_SPLITS"); if (e && atoi(e) > 0) s = atoi(e); }
  *tile_out = tile;
  return s;
}

// host side of the deferred combines.  One training step at a time; the autograd worker thread queues, the thread that
// drives the step begins and flushes -- hence a mutex, not thread_local state.
static std::mutex g_dw_group_mu;
static bool g_dw_group_on = false;
static DwReduceTable g_dw_group = {};

// GEMMs deferred as well (nacf_dw_group_begin(1)): what nacf_linear_bwd_weight was asked to do, replayed at the flush
struct DwGemmItem {
  const float* dZ; const float* X; float* dW; float* db; float* ws;
  int64_t lddz, ldx, lddw;
  size_t ws_bytes;
  int M, N, K, has_rs;
  float beta;
  nacf_rowset rs;
};
static std::vector<DwGemmItem> g_dw_items;
static int g_dw_defer_gemm = 0;
static int g_dw_last_group_launches = 0, g_dw_last_group_wgs = 0;

static int dw_combine_push_locked(float* slabs, float* dW, int64_t lddw, const float* part, float* db, int N, int K, int real_splits,
                                  float beta, hipStream_t s, int part_rows = -1);
static int dw_group_flush_locked(hipStream_t s);

// Launch every queued weight-gradient GEMM in grouped grids.  Splits: with W = sum of (output tiles x k-tiles) over the
// group, a workgroup should walk about W / target k-tiles (target = NACF_DW_GROUP_WGS, default 1024 = two rounds of the
// 512 resident 128x128 workgroups), never fewer than 24; a problem gets ceil(its k-tiles / that) splits.  Longest walks
// first in the grid, so that the short ones fill the tail.
static int dw_subset_launch_locked(const std::vector<int>& subset, hipStream_t s);
static int dw_items_launch_locked(hipStream_t s) {
  const int n_all = (int)g_dw_items.size();
  g_dw_last_group_launches = 0;
  g_dw_last_group_wgs = 0;
  if (n_all == 0) return NACF_OK;
  // (round 3 also had a one-workgroup-per-CU member for this launch, 128 x 256 tiles on the wide kernel's geometry with both
  //  operands through the transposing stager: +23 % on one long problem, +-0 on the step's mix -- DESIGN.md section 4; removed)
  // the g256 body builds 32-bit request offsets as row * (ld * 4) + column part with a 24-bit multiply (gemm_g256w.hpp: stage): an
  // operand of 2^24 rows, a row pitch of 2^24 bytes or 4 GiB in all would wrap -- such a group takes the 128 x 128 grouped kernel
  bool g256_fits = true;
  for (const DwGemmItem& it : g_dw_items) {
    const uint64_t ra = (uint64_t)it.lddz * 4u, rb = (uint64_t)it.ldx * 4u;
    if ((uint64_t)it.M >= (1u << 24) || ra >= (1u << 24) || rb >= (1u << 24) || (uint64_t)it.M * ra >= 0xfffffff0ull ||
        (uint64_t)it.M * rb >= 0xfffffff0ull)
      g256_fits = false;
  }
  if (g256_fits && g256_dw_enabled(gemm_mode())) {
    // bf16 matrix cores: the 256 x 256 eight-phase body on the fp32 operands, live-row gather in its DMA (nacf_gemm_g256.hip)
    std::vector<G256DwItem> gi(n_all);
    for (int i = 0; i < n_all; ++i) {
      const DwGemmItem& it = g_dw_items[i];
      G256DwItem& x = gi[i];
      x.dZ = it.dZ; x.X = it.X; x.dW = it.dW; x.db = it.db; x.ws = it.ws; x.ws_bytes = it.ws_bytes; x.lddz = it.lddz; x.ldx = it.ldx;
      x.lddw = it.lddw; x.M = it.M; x.N = it.N; x.K = it.K; x.rows = it.has_rs ? it.rs.rows : nullptr; x.count = it.has_rs ? it.rs.count : nullptr;
      x.beta = it.beta;
    }
    std::vector<G256Reduce> red;
    int wgs = 0;
    const int ns = gemm_mode() == NACF_GEMM_BF16X3 ? 3 : 1;
    int rc = g256_dw_group_launch(gi.data(), n_all, ns, red, &wgs, s);
    g_dw_items.clear();
    if (rc != NACF_OK) return rc;
    g_last_was_bf16 = true;
    bf16_note_wide(ns == 3 ? "g256_dw_group_kernel<3>" : "g256_dw_group_kernel<1>");      // (rocprofv3's name of the kernel, up to its namespace)
    g_dw_last_group_launches = 1;
    g_dw_last_group_wgs = wgs;
    for (const G256Reduce& r : red) {
      rc = dw_combine_push_locked(r.slabs, r.dW, r.lddw, r.part, r.db, r.N, r.K, r.splits, r.beta, s, r.part_rows);
      if (rc != NACF_OK) return rc;
    }
    return NACF_OK;
  }
  std::vector<int> all(n_all);
  for (int i = 0; i < n_all; ++i) all[i] = i;
  const int rc = dw_subset_launch_locked(all, s);
  g_dw_items.clear();
  return rc;
}
static int dw_subset_launch_locked(const std::vector<int>& subset, hipStream_t s) {
  const int n = (int)subset.size();
  static const int target_env = [] { const char* e = getenv("NACF_DW_GROUP_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const int mode = gemm_mode();
  // measured on the NACF step (bench.py, B = 128): exact mode 1024: 2.91 ms, 1280: 2.89, 1536: 2.90, 3072: 2.93;
  // throughput mode (its k-tiles are 3x shorter, the fixed cost of a split weighs less) 1024: 2.30, 1536: 2.12, 4096: 2.08
  const int tile_cols = 128;
  const int target = target_env > 0 ? target_env : (mode == NACF_GEMM_BF16 ? 4096 : 1280);
  // NACF_DW_GROUP_ORDER=0: every split spreads its tiles over the 8 XCDs (round 2); default 1: see GemmGroup
  static const bool split_major_env = [] { const char* e = getenv("NACF_DW_GROUP_ORDER"); return !(e && atoi(e) == 0); }();
  const bool split_major = split_major_env;
  std::vector<int> kt(n), tiles(n), sp(n), order(n);
  long W = 0;
  for (int i = 0; i < n; ++i) {
    const DwGemmItem& it = g_dw_items[subset[i]];
    const int m_eff = it.has_rs ? (int)((long)it.M * 29 / 50) : it.M;
    kt[i] = cdiv(m_eff > 0 ? m_eff : 1, 32);
    tiles[i] = cdiv(it.N, 128) * cdiv(it.K, tile_cols);
    W += (long)kt[i] * tiles[i];
    order[i] = i;
  }
  long walk = (W + target - 1) / target;
  if (walk < 24) walk = 24;
  for (int i = 0; i < n; ++i) {
    const DwGemmItem& it = g_dw_items[subset[i]];
    const size_t fixed = (size_t)64 * it.N * sizeof(float) + 256;
    long max_s = it.ws_bytes > fixed ? (long)((it.ws_bytes - fixed) / ((size_t)it.N * it.K * sizeof(float))) : 1;
    if (max_s > 64) max_s = 64;
    if (max_s < 1) max_s = 1;
    long sps = (kt[i] + walk - 1) / walk;
    if (sps > max_s) sps = max_s;
    if (sps < 1) sps = 1;
    sp[i] = (int)sps;
  }
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    const long wa = (long)kt[a] * sp[b], wb = (long)kt[b] * sp[a];      // kt / sp, longest first
    return wa != wb ? wa > wb : a < b;
  });
  int done = 0;
  while (done < n) {
    if (g_dw_group.n + GEMM_GROUP_MAX > DW_GROUP_MAX) {
      // room for this chunk's combines: everything queued so far belongs to GEMMs that are already in the stream
      hipLaunchKernelGGL(dw_group_reduce_kernel, dim3(g_dw_group.block_end), dim3(256), 0, s, g_dw_group);
      g_dw_group.n = 0;
      g_dw_group.block_end = 0;
      NACF_LAUNCH_CHECK("nacf_dw_group_flush(combine, between chunks)");
    }
    GemmGroup<EpiStore> t = {};
    t.order = split_major ? 1 : 0;
    int wg = 0;
    for (; done < n && t.n < GEMM_GROUP_MAX; ++done) {
      const DwGemmItem& it = g_dw_items[subset[order[done]]];
      const int i = order[done];
      GemmShape g = {};
      g.Q = it.dZ; g.P = it.X; g.ldq = it.lddz; g.ldp = it.ldx; g.M = it.N; g.N = it.K; g.K = it.M;
      g.k_per_split = cdiv(cdiv(it.M, sp[i]), 32) * 32;
      set_rows(g, it.has_rs ? &it.rs : nullptr);
      g.tiles_m = cdiv(g.M, 128);
      g.tiles_n = cdiv(g.N, tile_cols);
      const int real = cdiv(it.M, g.k_per_split);
      if (split_major) {
        // an XCD's run of `run` consecutive tiles inside one split: as square as the tile grid allows (GemmShape::group_n)
        const long run = ((long)tiles[i] * real + 7) / 8;
        if (run < tiles[i]) {
          int gn = 1;
          while ((long)(gn + 1) * (gn + 1) <= run) ++gn;
          if (gn < g.tiles_n) g.group_n = gn;
        }
      }
      float* slabs = it.ws;
      float* part = slabs + (real > 1 ? (size_t)real * it.N * it.K : 0);
      EpiStore epi;
      if (real > 1) {
        epi.C = slabs; epi.ldc = it.K; epi.beta = 0.f; epi.slab_stride = (int64_t)it.N * it.K; epi.vec_out = (it.K % 4 == 0) ? 1 : 0;
      } else {
        epi.C = it.dW; epi.ldc = it.lddw; epi.beta = it.beta; epi.slab_stride = 0;
        epi.vec_out = ((it.lddw % 4 == 0) && aligned16(it.dW)) ? 1 : 0;
      }
      if (it.db) {
        if (real > 1) g.colsum_part = part;
        else { g.colsum_out = it.db; g.colsum_beta = it.beta; }
      }
      const int gx = split_major ? tiles[i] : (tiles[i] + 7) / 8 * 8;
      t.g[t.n] = g; t.e[t.n] = epi; t.gx[t.n] = gx; t.nz[t.n] = real; t.wg0[t.n] = wg;
      wg += split_major ? (gx * real + 7) / 8 * 8 : gx * real;
      ++t.n;
      if (real > 1) {
        const int rc = dw_combine_push_locked(slabs, it.dW, it.lddw, it.db ? part : nullptr, it.db, it.N, it.K, real, it.beta, s);
        if (rc != NACF_OK) return rc;
      }
    }
    t.wg0[t.n] = wg;
    launch_bf16_dw_group(t, mode, s);
    g_last_was_bf16 = true;
    NACF_LAUNCH_CHECK("nacf_dw_group_flush(grouped gemm)");
    ++g_dw_last_group_launches;
    g_dw_last_group_wgs += wg;
  }
  return NACF_OK;
}

static int dw_combine_push_locked(float* slabs, float* dW, int64_t lddw, const float* part, float* db, int N, int K, int real_splits,
                                  float beta, hipStream_t s, int part_rows) {
  const bool v4 = (K % 4 == 0) && (lddw % 4 == 0) && aligned16(dW) && aligned16(slabs);
  const int64_t total = (int64_t)N * (v4 ? K / 4 : K);
  if (g_dw_group.n == DW_GROUP_MAX) {
    // NOTE: only legal when every GEMM whose combine is queued has been launched -- true for the immediate-GEMM mode; the
    // deferred-GEMM flush queues at most GEMM_GROUP_MAX combines per grouped launch and launches before the next chunk
    hipLaunchKernelGGL(dw_group_reduce_kernel, dim3(g_dw_group.block_end), dim3(256), 0, s, g_dw_group);
    g_dw_group.n = 0;
    g_dw_group.block_end = 0;
    NACF_LAUNCH_CHECK("nacf_dw_group(combine, queue full)");
  }
  for (int i = 0; i < g_dw_group.n; ++i)
    NACF_CHECK(g_dw_group.d[i].dst != dW, NACF_EINVAL, "nacf_linear_bwd_weight: this dW already has a deferred combine pending (flush first)");
  DwReduceDesc& d = g_dw_group.d[g_dw_group.n++];
  d.slabs = slabs; d.dst = dW; d.part = part; d.db = db; d.slab_stride = (int64_t)N * K; d.ldd = lddw;
  d.splits = real_splits; d.rows = N; d.cols = K; d.vec = v4 ? 1 : 0; d.beta = beta; d.block0 = g_dw_group.block_end;
  d.part_rows = part_rows;
  const int64_t want = real_splits > 0 ? (total + 1023) / 1024 : (N + 255) / 256;          // ~4 elements (float4s) per thread
  g_dw_group.block_end += (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  return NACF_OK;
}

static int dw_group_flush_locked(hipStream_t s) {
  {
    const int rc = dw_items_launch_locked(s);
    if (rc != NACF_OK) return rc;
  }
  if (g_dw_group.n > 0) {
    hipLaunchKernelGGL(dw_group_reduce_kernel, dim3(g_dw_group.block_end), dim3(256), 0, s, g_dw_group);
    g_dw_group.n = 0;
    g_dw_group.block_end = 0;
    NACF_LAUNCH_CHECK("nacf_dw_group_flush");
  }
  return NACF_OK;
}

int nacf_dw_group_begin(int defer_gemm) {
  std::lock_guard<std::mutex> lk(g_dw_group_mu);
  NACF_CHECK(g_dw_group.n == 0 && g_dw_items.empty(), NACF_EINVAL,
             "nacf_dw_group_begin: %d combines / %d GEMMs of the previous group were never flushed", g_dw_group.n, (int)g_dw_items.size());
  g_dw_group_on = true;
  g_dw_defer_gemm = defer_gemm ? 1 : 0;
  return NACF_OK;
}
int nacf_dw_group_stats(int* launches, int* workgroups) {
  std::lock_guard<std::mutex> lk(g_dw_group_mu);
  if (launches) *launches = g_dw_last_group_launches;
  if (workgroups) *workgroups = g_dw_last_group_wgs;
  return NACF_OK;
}
int nacf_dw_group_flush(nacf_stream_t stream) {
  std::lock_guard<std::mutex> lk(g_dw_group_mu);
  g_dw_group_on = false;
  return dw_group_flush_locked(as_hip(stream));
}
// the queued GEMMs only (grouped grids); the group stays open and the combines stay queued for nacf_dw_group_flush.
// For callers that time the grouped GEMM launch by itself (bench.py's roofline of the real step's dominant kernel).
int nacf_dw_group_launch_gemms(nacf_stream_t stream) {
  std::lock_guard<std::mutex> lk(g_dw_group_mu);
  const int n = (int)g_dw_items.size();
  const int rc = dw_items_launch_locked(as_hip(stream));
  return rc != NACF_OK ? rc : n;
}
int nacf_dw_group_pending(void) {
  std::lock_guard<std::mutex> lk(g_dw_group_mu);
  return g_dw_group_on ? g_dw_group.n + (int)g_dw_items.size() : -1;
}

size_t nacf_linear_bwd_weight_workspace(int M, int N, int K) {
  int tile;
  const int a = bwd_weight_splits(M, N, K, false, &tile), b = bwd_weight_splits(M, N, K, true, &tile);
  const int c = bwd_weight_splits_bf16(M, N, K, false, &tile), d = bwd_weight_splits_bf16(M, N, K, true, &tile);
  int s = a > b ? a : b;      // whichever kernel family the call ends up on (mode, alignment)
  if (c > s) s = c;
  if (d > s) s = d;
  const size_t slabs = s > 1 ? (size_t)s * N * K * sizeof(float) : 0;
  const size_t col = (size_t)64 * N * sizeof(float);
  const size_t g2 = g256_dw_enabled(gemm_mode()) ? g256_dw_workspace(M, N, K) : 0;      // (nacf_gemm_g256.hip)
  return slabs + col + 256 > g2 ? slabs + col + 256 : g2;
}

int nacf_linear_bwd_weight(const float* dZ, int64_t lddz, const float* X, int64_t ldx, float* dW, int64_t lddw,
                           float* db, int M, int N, int K, float beta, void* ws, size_t ws_bytes,
                           const nacf_rowset* rs, nacf_stream_t stream) {
  NACF_CHECK(dZ && X && dW, NACF_EINVAL, "nacf_linear_bwd_weight: null pointer");
  NACF_CHECK(M > 0 && N > 0 && K > 0, NACF_EINVAL, "nacf_linear_bwd_weight: bad shape");
  NACF_CHECK(lddz >= N && ldx >= K && lddw >= K, NACF_EINVAL, "nacf_linear_bwd_weight: leading dimension too small");
  NACF_CHECK(ws && ws_bytes >= nacf_linear_bwd_weight_workspace(M, N, K), NACF_EWORKSPACE,
             "nacf_linear_bwd_weight: workspace too small (%zu < %zu)", ws_bytes,
             nacf_linear_bwd_weight_workspace(M, N, K));
  NACF_CHECK(aligned16(ws), NACF_EINVAL, "nacf_linear_bwd_weight: workspace must be 16-byte aligned");
  NACF_CHECK(!rs || (rs->rows && rs->count), NACF_EINVAL, "nacf_linear_bwd_weight: incomplete row set");
  int tile;
  const bool vec = (lddz % 4 == 0) && (ldx % 4 == 0) && aligned16(dZ) && aligned16(X);
  const int mode = gemm_mode();
  const bool bf16 = mode != NACF_GEMM_F32 && vec;
  if (N < 64 && M <= 1024 && (int64_t)N * K <= (1 << 16) && forced_tile() < 0) {      // (the length head: models/Predictor.py:15-20)
    hipLaunchKernelGGL(dw_small_kernel, dim3(cdiv(N * K, 256)), dim3(256), 0, as_hip(stream), dZ, lddz, X, ldx, dW, lddw, db, M, N, K,
                       beta, rs ? rs->rows : nullptr, rs ? rs->count : nullptr);
    g_last_was_bf16 = false;
    snprintf(g_last_f32_kernel, sizeof(g_last_f32_kernel), "dw_small_kernel");
    NACF_LAUNCH_CHECK("nacf_linear_bwd_weight(small)");
    return NACF_OK;
  }
  const int splits = bf16 ? bwd_weight_splits_bf16(M, N, K, rs != nullptr, &tile) : bwd_weight_splits(M, N, K, rs != nullptr, &tile);
  hipStream_t s = as_hip(stream);
  if (bf16 && N >= 128 && K >= 128 && forced_tile() != 1) {
    std::lock_guard<std::mutex> lk(g_dw_group_mu);
    if (g_dw_group_on && g_dw_defer_gemm) {
      // the whole GEMM waits for the flush (grouped launch): dZ / X / the row set / ws stay untouched until then
      for (const DwGemmItem& it : g_dw_items)
        NACF_CHECK(it.dW != dW, NACF_EINVAL, "nacf_linear_bwd_weight: this dW already has a deferred GEMM pending (flush first)");
      for (int i = 0; i < g_dw_group.n; ++i)
        NACF_CHECK(g_dw_group.d[i].dst != dW, NACF_EINVAL, "nacf_linear_bwd_weight: this dW already has a deferred combine pending (flush first)");
      DwGemmItem it = {};
      it.dZ = dZ; it.X = X; it.dW = dW; it.db = db; it.ws = reinterpret_cast<float*>(ws); it.lddz = lddz; it.ldx = ldx; it.lddw = lddw;
      it.ws_bytes = ws_bytes; it.M = M; it.N = N; it.K = K; it.has_rs = rs ? 1 : 0; it.beta = beta;
      if (rs) it.rs = *rs;
      g_dw_items.push_back(it);
      return NACF_OK;
    }
  }
  // dW[n][k] = sum_m dZ[m][n] X[m][k]: output rows = n (Q = dZ, MC: element (n, m) at dZ[m*lddz + n]),
  // output cols = k (P = X, MC: element (k, m) at X[m*ldx + k]), reduce = m (through the live-row list if given)
  GemmShape g = {};
  g.Q = dZ; g.P = X; g.ldq = lddz; g.ldp = ldx; g.M = N; g.N = K; g.K = M;
  const int ka = bf16 ? 32 : 16;
  g.k_per_split = cdiv(cdiv(M, splits), ka) * ka;
  set_rows(g, rs);
  const int real_splits = cdiv(M, g.k_per_split);
  float* slabs = reinterpret_cast<float*>(ws);
  EpiStore epi;
  if (real_splits > 1) {
    epi.C = slabs; epi.ldc = K; epi.beta = 0.f; epi.slab_stride = (int64_t)N * K; epi.vec_out = (K % 4 == 0) ? 1 : 0;
  } else {
    epi.C = dW; epi.ldc = lddw; epi.beta = beta; epi.slab_stride = 0;
    epi.vec_out = ((lddw % 4 == 0) && aligned16(dW)) ? 1 : 0;
  }
  float* part = slabs + (splits > 1 ? (size_t)splits * N * K : 0);   // [real_splits][N] bias-gradient partials
  if (db) {
    if (real_splits > 1) g.colsum_part = part;
    else { g.colsum_out = db; g.colsum_beta = beta; }
  }
  if (bf16) {
    launch_bf16_dw(g, epi, real_splits, tile, mode, s);
    g_last_was_bf16 = true;
  } else {
    launch_gemm<false, false, EpiStore>(g, epi, real_splits, tile, vec, s);
  }
  NACF_LAUNCH_CHECK("nacf_linear_bwd_weight(gemm)");
  if (real_splits > 1) {
    const bool v4 = (K % 4 == 0) && (lddw % 4 == 0) && aligned16(dW) && aligned16(slabs);
    const int64_t total = (int64_t)N * (v4 ? K / 4 : K);
    {
      std::lock_guard<std::mutex> lk(g_dw_group_mu);
      // deferred: the caller keeps `ws` untouched until nacf_dw_group_flush and never queues one dW twice
      if (g_dw_group_on) return dw_combine_push_locked(slabs, dW, lddw, db ? part : nullptr, db, N, K, real_splits, beta, s);
    }
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (v4)
      hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, slabs, (int64_t)N * K, real_splits, dW,
                         lddw, N, K, beta, db ? part : nullptr, db);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, slabs, (int64_t)N * K, real_splits, dW,
                         lddw, N, K, beta, db ? part : nullptr, db);
    NACF_LAUNCH_CHECK("nacf_linear_bwd_weight(reduce)");
  }
  return NACF_OK;
}

#ifdef NACF_GEMM_TRACE
int nacf_debug_gemm_trace(void* buf) {     // tuning builds only (make trace); not part of the shipped ABI
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? NACF_OK : NACF_ELAUNCH;
}
#endif

int nacf_gemm_set_mode(int mode) {
  NACF_CHECK(mode == NACF_GEMM_F32 || mode == NACF_GEMM_BF16 || mode == NACF_GEMM_BF16X3, NACF_EINVAL,
             "nacf_gemm_set_mode: mode must be 0 (f32), 1 (bf16) or 3 (bf16x3), got %d", mode);
  g_mode = mode;
  return NACF_OK;
}
int nacf_gemm_get_mode(void) { return gemm_mode(); }
const char* nacf_gemm_last_kernel(void) { return g_last_was_bf16 ? bf16_last_kernel_name() : g_last_f32_kernel; }

int nacf_wimage_register(const float* w, int N, int K, int64_t ldw, const uint16_t* img, int64_t plane_elems,
                         const uint16_t* imgT, int64_t planeT_elems, int ns) {
  NACF_CHECK(w && N > 0 && K > 0 && ldw >= K && (img || imgT) && (ns == 1 || ns == 3), NACF_EINVAL,
             "nacf_wimage_register: bad argument");
  NACF_CHECK((!img || (aligned16(img) && plane_elems % 8 == 0 && plane_elems >= (int64_t)cdiv(K, 32) * 32 * N)) &&
             (!imgT || (aligned16(imgT) && planeT_elems % 8 == 0 && planeT_elems >= (int64_t)cdiv(N, 32) * 32 * K)),
             NACF_EINVAL, "nacf_wimage_register: images must be 16-byte aligned and hold whole 32-wide k-tiles");
  const ImgMat e{w, N, K, ldw, img, plane_elems, imgT, planeT_elems, ns};
  for (ImgMat& m : g_imgs)
    if (m.w == w && m.ns == ns) { m = e; return NACF_OK; }
  g_imgs.push_back(e);
  return NACF_OK;
}
int nacf_wimage_unregister(const float* w_base, int64_t n_elems) {
  NACF_CHECK(w_base && n_elems > 0, NACF_EINVAL, "nacf_wimage_unregister: bad argument");
  for (size_t i = g_imgs.size(); i-- > 0;)
    if (g_imgs[i].w >= w_base && g_imgs[i].w < w_base + n_elems) g_imgs.erase(g_imgs.begin() + i);
  return NACF_OK;
}
int nacf_wimage_refresh(const nacf_wimage_desc* table, int n_desc, int n_tiles, int ns, nacf_stream_t stream) {
  NACF_CHECK(table && n_desc > 0 && n_tiles > 0 && (ns == 1 || ns == 3), NACF_EINVAL, "nacf_wimage_refresh: bad argument");
  launch_wimage_refresh(table, n_desc, n_tiles, ns, as_hip(stream));
  NACF_LAUNCH_CHECK("nacf_wimage_refresh");
  return NACF_OK;
}

int nacf_gemm_config(int kind, int M, int N, int K, int* tile, int* splits) {
  NACF_CHECK(tile && splits && M > 0 && N > 0 && K > 0, NACF_EINVAL, "nacf_gemm_config: bad argument");
  int t = 0, s = 1;
  const bool with_rows = (kind & 0x100) != 0, heavy = (kind & 0x200) != 0;   // as the launch itself will see them
  kind &= 0xff;
  const int mode = gemm_mode();
  const bool bf16 = mode != NACF_GEMM_F32;     // (assumes 16-byte addressable operands)
  if (kind == 0) t = bf16 ? pick_tile_bf16(0, M, N, 1, with_rows, mode, heavy) : pick_tile(M, N, 1, with_rows, heavy);
  else if (kind == 1) {
    s = bwd_data_splits(M, N, K, with_rows);
    const int kps = cdiv(cdiv(N, s), kalign()) * kalign();
    s = cdiv(N, kps);
    t = bf16 ? pick_tile_bf16(1, M, K, s, with_rows, mode) : pick_tile(M, K, s, with_rows);
  }
  else if (kind == 2) {
    s = bf16 ? bwd_weight_splits_bf16(M, N, K, with_rows, &t) : bwd_weight_splits(M, N, K, with_rows, &t);
    const int kps = cdiv(cdiv(M, s), kalign()) * kalign();
    s = cdiv(M, kps);
  } else {
    nacf_set_error("nacf_gemm_config: bad kind %d", kind);
    return NACF_EINVAL;
  }
  *tile = t == 0 ? 128 : 64;
  *splits = s;
  return NACF_OK;
}

size_t nacf_vocab_argmax_workspace(int rows, int V) {
  const size_t tiles_n = (size_t)cdiv(V, 64);  // upper bound for either tile size
  return tiles_n * (size_t)rows * 3 * sizeof(float) + 256;
}

int nacf_vocab_argmax(const float* hidden, int64_t ldh, const float* W, int64_t ldw, const float* bias,
                      int rows, int V, int K, const int64_t* pad_tokens, int zero_mask_prob,
                      const uint8_t* update_mask, int64_t* tokens, float* probs, void* ws, size_t ws_bytes,
                      const nacf_rowset* rs, nacf_stream_t stream) {
  NACF_CHECK(hidden && W && tokens && probs, NACF_EINVAL, "nacf_vocab_argmax: null pointer");
  NACF_CHECK(rows > 0 && V > 0 && K > 0, NACF_EINVAL, "nacf_vocab_argmax: bad shape");
  NACF_CHECK(ws && ws_bytes >= nacf_vocab_argmax_workspace(rows, V), NACF_EWORKSPACE,
             "nacf_vocab_argmax: workspace too small");
  NACF_CHECK(!rs || (rs->rows && rs->count), NACF_EINVAL, "nacf_vocab_argmax: incomplete row set");
  const bool bf16_fam = gemm_mode() != NACF_GEMM_F32 && (ldh % 4 == 0) && (ldw % 4 == 0) && aligned16(hidden) && aligned16(W);
  const int tile = bf16_fam ? pick_tile_bf16(0, rows, V, 1, rs != nullptr, gemm_mode()) : pick_tile(rows, V, 1, rs != nullptr);
  GemmShape g = {};
  g.Q = hidden; g.P = W; g.ldq = ldh; g.ldp = ldw; g.M = rows; g.N = V; g.K = K;
  g.k_per_split = cdiv(K, 32) * 32;
  set_rows(g, rs);
  const bool vec = (ldh % 4 == 0) && (ldw % 4 == 0) && aligned16(hidden) && aligned16(W);
  hipStream_t s = as_hip(stream);
  const int mode = gemm_mode();
  if (mode != NACF_GEMM_F32 && vec) find_image(W, ldw, V, K, mode, g);
  // the DMA-fed exact-mode kernel (gemm_dma128.hpp) emits its partials per 128-column tile
  const int d128 = (mode == NACF_GEMM_BF16X3 && vec) ? dma128_pick(g, 1, rs != nullptr, 3, 2) : 0;
  const int tn = cdiv(V, (d128 || tile == 0) ? 128 : 64);
  EpiArgmax epi;
  epi.bias = bias;
  epi.pmax = reinterpret_cast<float*>(ws);
  epi.psum = epi.pmax + (size_t)tn * rows;
  epi.pidx = reinterpret_cast<int*>(epi.psum + (size_t)tn * rows);
  if (mode != NACF_GEMM_F32 && vec) {
    if (d128) launch_dma128_argmax(g, epi, d128, s);
    else launch_bf16_argmax(g, epi, tile, mode, s);
    g_last_was_bf16 = true;
  } else {
    launch_gemm<true, true, EpiArgmax>(g, epi, 1, tile, vec, s);
  }
  NACF_LAUNCH_CHECK("nacf_vocab_argmax(gemm)");
  hipLaunchKernelGGL(argmax_merge_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, epi.pmax, epi.psum, epi.pidx, tn, rows,
                     rs ? rs->rows : nullptr, rs ? rs->count : nullptr, pad_tokens, zero_mask_prob, update_mask, tokens,
                     probs);
  NACF_LAUNCH_CHECK("nacf_vocab_argmax(merge)");
  return NACF_OK;
}

// logits = hidden W^T + bias for the live rows, stored, AND their log-sum-exp / arg-max / log p(label) without a
// second pass over the [rows, V] matrix: the GEMM epilogue emits (max, arg-max, sum-exp) per column tile while the
// accumulators are still in registers, a small merge kernel finishes the rows.  Replaces nacf_linear_fwd +
// nacf_vocab_logsoftmax_fwd for models/__init__.py:83 + F.log_softmax + NLLLoss (misc/crit.py:62-114); the
// backward pass is nacf_xent_bwd_lse on the raw logits.
int nacf_vocab_lse_fwd(const float* hidden, int64_t ldh, const float* W, int64_t ldw, const float* bias, int rows, int V,
                       int K, float* logits, int64_t ldl, const int64_t* labels, float* lse, int64_t* argmax,
                       float* label_logp, void* ws, size_t ws_bytes, const nacf_rowset* rs, nacf_stream_t stream) {
  NACF_CHECK(hidden && W && logits && lse, NACF_EINVAL, "nacf_vocab_lse_fwd: null pointer");
  NACF_CHECK(rows > 0 && V > 0 && K > 0 && ldl >= V, NACF_EINVAL, "nacf_vocab_lse_fwd: bad shape");
  NACF_CHECK(ldl % 4 == 0 && aligned16(logits), NACF_EINVAL, "nacf_vocab_lse_fwd: logits need 16-byte aligned rows (nacf vocab_ld)");
  NACF_CHECK(ws && ws_bytes >= nacf_vocab_argmax_workspace(rows, V), NACF_EWORKSPACE, "nacf_vocab_lse_fwd: workspace too small");
  NACF_CHECK(!rs || (rs->rows && rs->count), NACF_EINVAL, "nacf_vocab_lse_fwd: incomplete row set");
  const bool bf16_fam = gemm_mode() != NACF_GEMM_F32 && (ldh % 4 == 0) && (ldw % 4 == 0) && aligned16(hidden) && aligned16(W);
  const int tile = bf16_fam ? pick_tile_bf16(0, rows, V, 1, rs != nullptr, gemm_mode()) : pick_tile(rows, V, 1, rs != nullptr);
  GemmShape g = {};
  g.Q = hidden; g.P = W; g.ldq = ldh; g.ldp = ldw; g.M = rows; g.N = V; g.K = K;
  g.k_per_split = cdiv(K, 32) * 32;
  set_rows(g, rs);
  g.zero_dead = 0;                       // rows without a label are never read downstream
  const bool vec = (ldh % 4 == 0) && (ldw % 4 == 0) && aligned16(hidden) && aligned16(W);
  hipStream_t s = as_hip(stream);
  const int mode = gemm_mode();
  if (mode != NACF_GEMM_F32 && vec) find_image(W, ldw, V, K, mode, g);
  const int d128 = (mode == NACF_GEMM_BF16X3 && vec) ? dma128_pick(g, 1, rs != nullptr, 3, 2) : 0;
  const int tn = cdiv(V, (d128 || tile == 0) ? 128 : 64);
  EpiArgmax epi;
  epi.bias = bias;
  epi.pmax = reinterpret_cast<float*>(ws);
  epi.psum = epi.pmax + (size_t)tn * rows;
  epi.pidx = reinterpret_cast<int*>(epi.psum + (size_t)tn * rows);
  epi.C = logits;
  epi.ldc = ldl;
  if (mode != NACF_GEMM_F32 && vec) {
    if (d128) launch_dma128_argmax(g, epi, d128, s);
    else launch_bf16_argmax(g, epi, tile, mode, s);
    g_last_was_bf16 = true;
  } else {
    launch_gemm<true, true, EpiArgmax>(g, epi, 1, tile, vec, s);
  }
  NACF_LAUNCH_CHECK("nacf_vocab_lse_fwd(gemm)");
  hipLaunchKernelGGL(lse_merge_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, epi.pmax, epi.psum, epi.pidx, tn, rows,
                     rs ? rs->rows : nullptr, rs ? rs->count : nullptr, logits, ldl, labels, lse, argmax, label_logp);
  NACF_LAUNCH_CHECK("nacf_vocab_lse_fwd(merge)");
  return NACF_OK;
}

}  // extern "C"

// Decoder-side kernels: BertEmbeddings (gather + LayerNorm), its backward and
// the deterministic scatter into the embedding tables, the multi-head
// attention core (forward + backward) and the masked row mean.
//
// Attention here is tiny (Lq <= 30, Lk <= 120, dk = 64: <1% of the step's
// FLOPs) so it runs on the fp32 VALU out of LDS; one workgroup
// per (sequence, head) keeps K/V/scores on chip for the whole softmax(QK^T)V.
#include "common.hpp"
#include "attn_mfma.hpp"
#include <stdlib.h>

namespace {

// shapes the register-resident MFMA attention covers; anything else (or NACF_ATTN_VALU=1)
// takes the generic LDS/VALU kernels below
inline bool attn_mfma_ok(int Lq, int Lk, int dk) {
  const char* e = getenv("NACF_ATTN_VALU");
  if (e && atoi(e) != 0) return false;
  return Lq <= 32 && Lk <= 128 && (dk == 16 || dk == 64);
}
// arithmetic of the matrix-core attention (attn_mfma.hpp "PR"): bf16 operands in the throughput mode of the GEMMs, at the model's
// head width; NACF_ATTN_BF16=0 | 1 overrides (A/B, tests)
inline int attn_precision(int dk) {
  if (dk != 64) return 0;
  const char* e = getenv("NACF_ATTN_BF16");
  if (e && *e) return atoi(e) != 0 ? 1 : 0;
  return nacf_gemm_get_mode() == NACF_GEMM_BF16 ? 1 : 0;
}
inline bool attn_aligned(const void* p, int64_t ld) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % 4) == 0;
}

constexpr int EMB_THREADS = 128;
constexpr int EMB_MAXJ = 4;  // D <= 4 * 128 * 4 = 2048

__global__ __launch_bounds__(EMB_THREADS) void embed_ln_fwd_kernel(
    const int64_t* __restrict__ tokens, const int64_t* __restrict__ category, const float* __restrict__ additional,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ cat,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* __restrict__ out,
    float* __restrict__ xhat, float* __restrict__ rstd_out, int L, int D, int vdiv, int vmod, float eps, float p,
    uint32_t salt, const uint64_t* __restrict__ rng_state) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int r = row / L, l = row % L;
  const int v = (r / vdiv) % vmod;
  const int64_t tok = tokens[row];
  const int64_t c = (cat && category) ? category[v] : 0;
  f32x4 x[EMB_MAXJ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    x[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (d < D) {
      f32x4 a = *reinterpret_cast<const f32x4*>(word + tok * D + d);
      f32x4 b = *reinterpret_cast<const f32x4*>(pos + (int64_t)l * D + d);
      a += b;
      if (cat && category) a += *reinterpret_cast<const f32x4*>(cat + c * D + d);
      if (additional) a += *reinterpret_cast<const f32x4*>(additional + (int64_t)v * D + d);
      x[j] = a;
      sum += a[0] + a[1] + a[2] + a[3];
    }
  }
  const float mean = block_sum(sum, red) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dv = x[j][e] - mean; sq += dv * dv; }
    }
  }
  const float var = block_sum(sq, red) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
      f32x4 xh;
#pragma unroll
      for (int e = 0; e < 4; ++e) xh[e] = (x[j][e] - mean) * rstd;
      if (xhat) *reinterpret_cast<f32x4*>(xhat + (int64_t)row * D + d) = xh;
      const f32x4 w = *reinterpret_cast<const f32x4*>(ln_w + d);
      const f32x4 b = *reinterpret_cast<const f32x4*>(ln_b + d);
      f32x4 y = xh * w + b;
      if (p > 0.f) y *= rng.keep4(((uint64_t)row * D + d) >> 2, salt, p);
      *reinterpret_cast<f32x4*>(out + (int64_t)row * D + d) = y;
    }
  }
  if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
}

// ONE WAVE per row (D = 256 NJ, NJ <= 4: the model's 512), four rows per workgroup: the two LayerNorm reductions are wave
// shuffles instead of two workgroup barriers each -- a row's chain is token -> embedding rows -> mean -> variance -> store, and
// with one 128-thread workgroup per row (14592 of them for a decode canvas) that chain was barrier latency: 20 us per pass
// for 30 MB.  Same values up to the order of the two sums.
template <int NJ>
__global__ __launch_bounds__(256) void embed_ln_fwd_wave_kernel(
    const int64_t* __restrict__ tokens, const int64_t* __restrict__ category, const float* __restrict__ additional,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ cat,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* __restrict__ out,
    float* __restrict__ xhat, float* __restrict__ rstd_out, int rows, int L, int vdiv, int vmod, float eps, float p,
    uint32_t salt, const uint64_t* __restrict__ rng_state) {
  constexpr int D = 256 * NJ;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int r = row / L, l = row % L;
  const int v = (r / vdiv) % vmod;
  const int64_t tok = tokens[row];
  const bool has_cat = cat && category;
  const int64_t c = has_cat ? category[v] : 0;
  f32x4 x[NJ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int d = (lane + 64 * j) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(word + tok * D + d);
    a += *reinterpret_cast<const f32x4*>(pos + (int64_t)l * D + d);
    if (has_cat) a += *reinterpret_cast<const f32x4*>(cat + c * D + d);
    if (additional) a += *reinterpret_cast<const f32x4*>(additional + (int64_t)v * D + d);
    x[j] = a;
    sum += a[0] + a[1] + a[2] + a[3];
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float dv = x[j][e] - mean; sq += dv * dv; }
  const float var = wave_sum(sq) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int d = (lane + 64 * j) * 4;
    f32x4 xh;
#pragma unroll
    for (int e = 0; e < 4; ++e) xh[e] = (x[j][e] - mean) * rstd;
    if (xhat) *reinterpret_cast<f32x4*>(xhat + (int64_t)row * D + d) = xh;
    const f32x4 w = *reinterpret_cast<const f32x4*>(ln_w + d);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ln_b + d);
    f32x4 y = xh * w + b;
    if (p > 0.f) y *= rng.keep4(((uint64_t)row * D + d) >> 2, salt, p);
    *reinterpret_cast<f32x4*>(out + (int64_t)row * D + d) = y;
  }
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
}

__global__ __launch_bounds__(EMB_THREADS) void embed_ln_bwd_kernel(
    const float* __restrict__ dOut, const float* __restrict__ xhat, const float* __restrict__ rstd,
    const float* __restrict__ ln_w, float* __restrict__ dE, float* __restrict__ part, int rows, int D, float p,
    uint32_t salt, const uint64_t* __restrict__ rng_state) {
  __shared__ float red[16];
  f32x4 dw[EMB_MAXJ], db[EMB_MAXJ];
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) { dw[j] = f32x4{0.f, 0.f, 0.f, 0.f}; db[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
  const float invD = 1.f / (float)D;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    f32x4 dxh[EMB_MAXJ], xh[EMB_MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (threadIdx.x + EMB_THREADS * j) * 4;
      dxh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      xh[j] = dxh[j];
      if (d < D) {
        f32x4 dy = *reinterpret_cast<const f32x4*>(dOut + (int64_t)row * D + d);
        if (p > 0.f) dy *= rng.keep4(((uint64_t)row * D + d) >> 2, salt, p);
        xh[j] = *reinterpret_cast<const f32x4*>(xhat + (int64_t)row * D + d);
        const f32x4 w = *reinterpret_cast<const f32x4*>(ln_w + d);
        dw[j] += dy * xh[j];
        db[j] += dy;
        dxh[j] = dy * w;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1 += dxh[j][e]; s2 += dxh[j][e] * xh[j][e]; }
      }
    }
    s1 = block_sum(s1, red) * invD;
    s2 = block_sum(s2, red) * invD;
    const float rs = rstd[row];
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (threadIdx.x + EMB_THREADS * j) * 4;
      if (d < D) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (dxh[j][e] - s1 - xh[j][e] * s2);
        *reinterpret_cast<f32x4*>(dE + (int64_t)row * D + d) = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
      *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 0) * D + d) = dw[j];
      *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 1) * D + d) = db[j];
    }
  }
}
// combine the per-block LayerNorm weight / bias gradient partials part[k][{w,b}][D]: a workgroup owns 16 columns and
// splits the k walk over 16 groups (group gq sums k = gq, gq+16, ...), then folds the groups in fixed order
__global__ __launch_bounds__(256) void embed_ln_bwd_final_kernel(const float* __restrict__ part, int nblk, int D,
                                                                 float* __restrict__ dln_w, float* __restrict__ dln_b, float beta) {
  __shared__ float red[2][16][17];
  const int c = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int d = blockIdx.x * 16 + c;
  float a = 0.f, b = 0.f;
  if (d < D) {
#pragma unroll 4
    for (int k = gq; k < nblk; k += 16) {
      a += part[((int64_t)k * 2 + 0) * D + d];
      b += part[((int64_t)k * 2 + 1) * D + d];
    }
  }
  red[0][gq][c] = a;
  red[1][gq][c] = b;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int which = threadIdx.x >> 4;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[which][i][c];
    float* dst = which ? dln_b : dln_w;
    if (dst && d < D) dst[d] = (beta != 0.f) ? t + beta * dst[d] : t;
  }
}

// ------------------------------------------------------------------ generic LayerNorm (with_layernorm / norm_type=ln)
// out[orow] = mask(dropout(LN(x[r]))),  orow = (r / seg_in) * seg_out + seg_off + r % seg_in
__global__ __launch_bounds__(EMB_THREADS) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* __restrict__ out,
    float* __restrict__ xhat, float* __restrict__ rstd_out, int D, float eps, int seg_in, int seg_out, int seg_off,
    float p, uint32_t salt, const uint64_t* __restrict__ rng_state, const int64_t* __restrict__ row_tokens) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int64_t orow = (int64_t)(row / seg_in) * seg_out + seg_off + row % seg_in;
  f32x4 v[EMB_MAXJ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (d < D) {
      v[j] = *reinterpret_cast<const f32x4*>(x + (int64_t)row * D + d);
      sum += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
  }
  const float mean = block_sum(sum, red) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dv = v[j][e] - mean; sq += dv * dv; }
    }
  }
  const float rstd = 1.0f / sqrtf(block_sum(sq, red) / (float)D + eps);
  const bool dead = row_tokens && row_tokens[row] == NACF_PAD;
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
      f32x4 xh;
#pragma unroll
      for (int e = 0; e < 4; ++e) xh[e] = (v[j][e] - mean) * rstd;
      if (xhat) *reinterpret_cast<f32x4*>(xhat + (int64_t)row * D + d) = xh;
      f32x4 y = xh * *reinterpret_cast<const f32x4*>(ln_w + d) + *reinterpret_cast<const f32x4*>(ln_b + d);
      if (p > 0.f) y *= rng.keep4(((uint64_t)row * D + d) >> 2, salt, p);
      if (dead) y = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(out + orow * D + d) = y;
    }
  }
  if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
}

__global__ __launch_bounds__(EMB_THREADS) void layernorm_bwd_kernel(
    const float* __restrict__ dOut, const float* __restrict__ xhat, const float* __restrict__ rstd,
    const float* __restrict__ ln_w, float* __restrict__ dX, float* __restrict__ part, int rows, int D, int seg_in,
    int seg_out, int seg_off, float p, uint32_t salt, const uint64_t* __restrict__ rng_state,
    const int64_t* __restrict__ row_tokens) {
  __shared__ float red[16];
  f32x4 dw[EMB_MAXJ], db[EMB_MAXJ];
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) { dw[j] = f32x4{0.f, 0.f, 0.f, 0.f}; db[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
  const float invD = 1.f / (float)D;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int64_t orow = (int64_t)(row / seg_in) * seg_out + seg_off + row % seg_in;
    const bool dead = row_tokens && row_tokens[row] == NACF_PAD;
    f32x4 dxh[EMB_MAXJ], xh[EMB_MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (threadIdx.x + EMB_THREADS * j) * 4;
      dxh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      xh[j] = dxh[j];
      if (d < D) {
        f32x4 dy = dead ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(dOut + orow * D + d);
        if (p > 0.f) dy *= rng.keep4(((uint64_t)row * D + d) >> 2, salt, p);
        xh[j] = *reinterpret_cast<const f32x4*>(xhat + (int64_t)row * D + d);
        dw[j] += dy * xh[j];
        db[j] += dy;
        dxh[j] = dy * *reinterpret_cast<const f32x4*>(ln_w + d);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1 += dxh[j][e]; s2 += dxh[j][e] * xh[j][e]; }
      }
    }
    s1 = block_sum(s1, red) * invD;
    s2 = block_sum(s2, red) * invD;
    const float rs = rstd[row];
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (threadIdx.x + EMB_THREADS * j) * 4;
      if (d < D) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (dxh[j][e] - s1 - xh[j][e] * s2);
        *reinterpret_cast<f32x4*>(dX + (int64_t)row * D + d) = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
      *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 0) * D + d) = dw[j];
      *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * 2 + 1) * D + d) = db[j];
    }
  }
}

// ---- embedding-table gradients: two launches.  Phase A runs the four independent gathers side by side (block ranges
// of ONE grid: each is a chain of dependent row loads, together they cost what the longest does), phase B the three
// fixed-order combines.  Every sum keeps the order of the one-kernel-per-table version (ascending rows / chunks / videos).
constexpr int SCAT_GROUPS = 4;
constexpr int SCAT_THREADS = SCAT_GROUPS * EMB_THREADS;      // 512: 8 waves
constexpr int SPECIAL_N = 5;          // ids 1..5: <unk>, <bos>, <eos>, <mask>, <vis>
constexpr int SCATTER_CHUNK = 128;
struct ScatShared {
  int list[SCATTER_CHUNK];
  int wcnt[2];
  f32x4 red[SCAT_GROUPS][EMB_MAXJ][EMB_THREADS];
};

// dword[tok] += sum of the dE rows carrying `tok`, in ascending row order; the WAVE of the FIRST occurrence of a token
// does the whole sum (one wave per row, no workgroup barrier: 8 rows per workgroup)
__device__ __forceinline__ void scatter_word_wave(const float* __restrict__ dE, const int64_t* __restrict__ tokens,
                                                  float* __restrict__ dword, int n_rows, int D, int first_tok, int me) {
  if (me >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t tok = tokens[me];
  if (tok < first_tok) return;  // PAD never gets a gradient; ids < first_tok go through the chunked path
  // both scans walk the token list 256 entries per trip (four independent loads in flight: one load + ballot per trip
  // is a chain of ~80 dependent L2 round trips per row, 32 us of the launch)
  for (int base = 0; base < me; base += 256) {
    int64_t t4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t4[u] = tokens[min(base + 64 * u + lane, n_rows - 1)];      // unconditional: all four in flight
    bool hit = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) hit |= (base + 64 * u + lane < me) & (t4[u] == tok);
    if (__ballot(hit) != 0ull) return;
  }
  constexpr int WJ = 2 * EMB_MAXJ;      // float4 columns per lane
  f32x4 acc[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = me; base < n_rows; base += 256) {
    int64_t t4[4];
    bool m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t4[u] = tokens[min(base + 64 * u + lane, n_rows - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = (base + 64 * u + lane < n_rows) & (t4[u] == tok);
#pragma unroll
    for (int u = 0; u < 4; ++u) {          // ascending rows: chunk by chunk, bit by bit
      unsigned long long bal = __ballot(m[u]);
      while (bal) {
        const int bit = __builtin_ctzll(bal);
        bal &= bal - 1ull;
        const int64_t src = (int64_t)(base + 64 * u + bit) * D;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          const int d = (lane + 64 * j) * 4;
          if (d < D) acc[j] += *reinterpret_cast<const f32x4*>(dE + src + d);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int d = (lane + 64 * j) * 4;
    if (d < D) {
      float* o = dword + tok * D + d;
      const f32x4 cur = *reinterpret_cast<f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = cur + acc[j];
    }
  }
}

// The special ids label thousands of rows each (every masked / visual-word slot), so their rows are summed
// chunk-parallel: partial[chunk][tok-1][D] over 128-row chunks, then a fixed-order combine.  The first EMB_THREADS
// threads list the chunk's rows that hold this token (ascending); group gq then adds list entries gq, gq+4, ... and
// the 4 partials fold through LDS in fixed order
__device__ __forceinline__ void scatter_special_partial(ScatShared& sh, const float* __restrict__ dE,
                                                        const int64_t* __restrict__ tokens, float* __restrict__ part, int n_rows,
                                                        int D, int chunk, int ti) {
  const int64_t tok = ti + 1;
  const int t = threadIdx.x % EMB_THREADS, gq = threadIdx.x / EMB_THREADS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool match = false;
  int pre = 0;
  if (gq == 0) {
    const int idx = chunk * SCATTER_CHUNK + t;
    match = idx < n_rows && tokens[idx] == tok;
    const unsigned long long bal = __ballot(match);
    pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) sh.wcnt[wave] = __popcll(bal);
  }
  __syncthreads();
  const int n = sh.wcnt[0] + sh.wcnt[1];
  if (gq == 0 && match) sh.list[((wave == 0) ? 0 : sh.wcnt[0]) + pre] = chunk * SCATTER_CHUNK + t;
  __syncthreads();
  f32x4 acc[EMB_MAXJ];
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = gq; i < n; i += SCAT_GROUPS) {
    const int64_t src = (int64_t)sh.list[i] * D;
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (t + EMB_THREADS * j) * 4;
      if (d < D) acc[j] += *reinterpret_cast<const f32x4*>(dE + src + d);
    }
  }
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) sh.red[gq][j][t] = acc[j];
  __syncthreads();
  if (gq == 0) {
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (t + EMB_THREADS * j) * 4;
      if (d < D)
        *reinterpret_cast<f32x4*>(part + ((int64_t)chunk * SPECIAL_N + ti) * D + d) =
            ((sh.red[0][j][t] + sh.red[1][j][t]) + sh.red[2][j][t]) + sh.red[3][j][t];
    }
  }
}

// part[pc][l][D] = sum over the rows r of chunk pc of dE[r, l]   (the first EMB_THREADS threads of the workgroup)
__device__ __forceinline__ void scatter_pos_partial(const float* __restrict__ dE, float* __restrict__ part, int R, int L, int D,
                                                    int rows_per, int l, int pc) {
  if (threadIdx.x >= EMB_THREADS) return;
  const int r0 = pc * rows_per, r1 = min(R, r0 + rows_per);
  f32x4 acc[EMB_MAXJ];
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < r1; ++r) {
    const int64_t src = ((int64_t)r * L + l) * D;
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (threadIdx.x + EMB_THREADS * j) * 4;
      if (d < D) acc[j] += *reinterpret_cast<const f32x4*>(dE + src + d);
    }
  }
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) *reinterpret_cast<f32x4*>(part + ((int64_t)pc * L + l) * D + d) = acc[j];
  }
}

// vsum[v] = sum over the decoder rows of video v and all positions: position group gq walks pairs gq, gq+4, ... of the
// (row, position) pairs of the video; the 4 partials fold through LDS in fixed order
__device__ __forceinline__ void scatter_video(ScatShared& sh, const float* __restrict__ dE, float* __restrict__ vsum, int R, int L,
                                              int D, int vdiv, int vmod, int me) {
  const int t = threadIdx.x % EMB_THREADS, gq = threadIdx.x / EMB_THREADS;
  f32x4 acc[EMB_MAXJ];
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the rows of video `me` are r = (q * vmod + me) * vdiv + j  (j < vdiv, q = 0, 1, ...)
  const int per_q = vdiv * L;
  const int n_q = (R / vdiv - me + vmod - 1) / vmod;          // number of q with (q*vmod + me) < R/vdiv
  const int n_pairs = (n_q > 0 ? n_q : 0) * per_q;
#pragma unroll 2
  for (int pidx = gq; pidx < n_pairs; pidx += SCAT_GROUPS) {
    const int q = pidx / per_q, rem = pidx % per_q;
    const int r = (q * vmod + me) * vdiv + rem / L, l = rem % L;
    const int64_t src = ((int64_t)r * L + l) * D;
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (t + EMB_THREADS * j) * 4;
      if (d < D) acc[j] += *reinterpret_cast<const f32x4*>(dE + src + d);
    }
  }
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) sh.red[gq][j][t] = acc[j];
  __syncthreads();
  if (gq == 0) {
#pragma unroll
    for (int j = 0; j < EMB_MAXJ; ++j) {
      const int d = (t + EMB_THREADS * j) * 4;
      if (d < D) *reinterpret_cast<f32x4*>(vsum + (int64_t)me * D + d) = ((sh.red[0][j][t] + sh.red[1][j][t]) + sh.red[2][j][t]) + sh.red[3][j][t];
    }
  }
}

struct ScatterArgs {
  const float* dE; const int64_t* tokens; const int64_t* category;
  float* dword; float* dpos; float* dcat; float* vsum;
  float* part_special; float* part_pos;
  int R, L, D, rows, first_tok;
  int chunks, n_special;        // special ids: grid chunks x n_special
  int pos_real, pos_rows_per;   // positions: grid L x pos_real
  int n_video, vdiv, vmod, n_cat;
  int b_special, b_video, b_pos, b_word;     // phase A: first workgroup of each role (b_word .. grid end: 8 rows each)
  int c_special, c_pos, c_cat;               // phase B
};

__global__ __launch_bounds__(SCAT_THREADS) void embed_scatter_gather_kernel(ScatterArgs a) {
  __shared__ ScatShared sh;
  const int b = blockIdx.x;
  if (b >= a.b_word) {
    scatter_word_wave(a.dE, a.tokens, a.dword, a.rows, a.D, a.first_tok, (b - a.b_word) * (SCAT_THREADS / 64) + (threadIdx.x >> 6));
  } else if (b >= a.b_pos) {
    const int i = b - a.b_pos;
    scatter_pos_partial(a.dE, a.part_pos, a.R, a.L, a.D, a.pos_rows_per, i % a.L, i / a.L);
  } else if (b >= a.b_video) {
    scatter_video(sh, a.dE, a.vsum, a.R, a.L, a.D, a.vdiv, a.vmod, b - a.b_video);
  } else {
    const int i = b - a.b_special;
    scatter_special_partial(sh, a.dE, a.tokens, a.part_special, a.rows, a.D, i % a.chunks, i / a.chunks);
  }
}

// dst[(row0 + item)] (+)= sum_{c < n_parts} part[c][item]  (fixed order)
__device__ __forceinline__ void scatter_combine(const float* __restrict__ part, int n_parts, int n_items, float* __restrict__ dst,
                                                int row0, int D, int accumulate, int item) {
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int c = 0; c < n_parts; ++c) acc += *reinterpret_cast<const f32x4*>(part + ((int64_t)c * n_items + item) * D + d);
      float* o = dst + (int64_t)(row0 + item) * D + d;
      if (accumulate) acc += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = acc;
    }
  }
}
// dcat[c] += sum of vsum[v] over the videos of category c (ascending v)
__device__ __forceinline__ void scatter_cat(const float* __restrict__ vsum, const int64_t* __restrict__ category,
                                            float* __restrict__ dcat, int n_video, int D, int me) {
  f32x4 acc[EMB_MAXJ];
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the videos of this category, ascending: 128 category ids per trip (one load per thread, matches listed through LDS)
  // instead of one dependent load + branch per video
  __shared__ int vlist[EMB_THREADS];
  __shared__ int vcnt[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n_video; base += EMB_THREADS) {
    const int v = base + threadIdx.x;
    const bool match = v < n_video && category[v] == me;
    const unsigned long long bal = __ballot(match);
    if (lane == 0) vcnt[wave] = __popcll(bal);
    __syncthreads();
    if (match) vlist[(wave == 0 ? 0 : vcnt[0]) + __popcll(bal & ((1ull << lane) - 1ull))] = v;
    __syncthreads();
    const int n = vcnt[0] + vcnt[1];
    for (int i = 0; i < n; ++i) {
      const int64_t src = (int64_t)vlist[i] * D;
#pragma unroll
      for (int j = 0; j < EMB_MAXJ; ++j) {
        const int d = (threadIdx.x + EMB_THREADS * j) * 4;
        if (d < D) acc[j] += *reinterpret_cast<const f32x4*>(vsum + src + d);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < EMB_MAXJ; ++j) {
    const int d = (threadIdx.x + EMB_THREADS * j) * 4;
    if (d < D) {
      float* o = dcat + (int64_t)me * D + d;
      const f32x4 cur = *reinterpret_cast<f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = cur + acc[j];
    }
  }
}
__global__ __launch_bounds__(EMB_THREADS) void embed_scatter_combine_kernel(ScatterArgs a) {
  const int b = blockIdx.x;
  if (b >= a.c_cat) scatter_cat(a.vsum, a.category, a.dcat, a.n_video, a.D, b - a.c_cat);
  else if (b >= a.c_pos) scatter_combine(a.part_pos, a.pos_real, a.L, a.dpos, 0, a.D, 1, b - a.c_pos);
  // partial layout is [chunk][SPECIAL_N][D]: n_items = SPECIAL_N, only the first n_special used; dword rows 1..
  else scatter_combine(a.part_special, a.chunks, SPECIAL_N, a.dword, 1, a.D, 1, b - a.c_special);
}

__global__ void masked_mean_fwd_kernel(const float* __restrict__ y, const int64_t* __restrict__ tokens,
                                       float* __restrict__ out, int L, int D) {
  const int r = blockIdx.x;
  int cnt = 0;
#pragma unroll 8
  for (int l = 0; l < L; ++l) cnt += tokens[(int64_t)r * L + l] != NACF_PAD ? 1 : 0;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float acc = 0.f;
#pragma unroll 8
    for (int l = 0; l < L; ++l) acc += y[((int64_t)r * L + l) * D + d];
    out[(int64_t)r * D + d] = acc / (float)cnt;
  }
}

// ------------------------------------------------------------------ attention
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ Q, int64_t ldq,
                                                             const float* __restrict__ K, int64_t ldk,
                                                             const float* __restrict__ V, int64_t ldv,
                                                             float* __restrict__ O, int64_t ldo,
                                                             const int64_t* __restrict__ key_tokens, int causal,
                                                             float* __restrict__ probs, int R, int Lq, int Lk, int dk,
                                                             int kv_div, int kv_mod, float p_drop, uint32_t salt,
                                                             const uint64_t* __restrict__ rng_state) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int r = blockIdx.x, h = blockIdx.y;
  // attention_probs_dropout_prob (models/bert.py:135,169): element ((r * H + h) * Lq + q) * Lk + k of the device Philox stream
  DropRng rng;
  if (p_drop > 0.f) rng.init(rng_state);
  const uint64_t e0 = ((uint64_t)r * gridDim.y + h) * (uint64_t)Lq * (uint64_t)Lk;
  const int kvr = (r / kv_div) % kv_mod;
  const int ldd = dk + 1;
  float* qs = sm;                // [Lq][dk+1]
  float* ks = qs + Lq * ldd;     // [Lk][dk+1], later V as [Lk][dk]
  float* ps = ks + Lk * ldd;     // [Lq][Lk]
  const int tid = threadIdx.x;
  for (int idx = tid; idx < Lq * dk; idx += 256) {
    const int q = idx / dk, d = idx % dk;
    qs[q * ldd + d] = Q[((int64_t)r * Lq + q) * ldq + h * dk + d];
  }
  for (int idx = tid; idx < Lk * dk; idx += 256) {
    const int k = idx / dk, d = idx % dk;
    ks[k * ldd + d] = K[((int64_t)kvr * Lk + k) * ldk + h * dk + d];
  }
  __syncthreads();
  const float sq = sqrtf((float)dk);
  for (int idx = tid; idx < Lq * Lk; idx += 256) {
    const int q = idx / Lk, k = idx % Lk;
    float dot = 0.f;
    const float* a = qs + q * ldd;
    const float* b = ks + k * ldd;
    for (int d = 0; d < dk; ++d) dot += a[d] * b[d];
    float s = dot / sq;
    const bool masked = (key_tokens && key_tokens[(int64_t)r * Lk + k] == NACF_PAD) || (causal && (k > q || (causal > 1 && k <= q - (causal - 1))));
    if (masked) s = -10e6f;  // models/bert.py:161: -10e6, not -inf
    ps[idx] = s;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int q = wave; q < Lq; q += 4) {
    float* row = ps + q * Lk;
    float m = -3.0e38f;
    for (int k = lane; k < Lk; k += 64) m = fmaxf(m, row[k]);
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < Lk; k += 64) { const float e = expf(row[k] - m); row[k] = e; s += e; }
    s = wave_sum(s);
    for (int k = lane; k < Lk; k += 64) {
      float p = row[k] / s;
      if (p_drop > 0.f) p *= rng.keep1(e0 + (uint64_t)q * Lk + k, salt, p_drop);      // (the returned probabilities are the dropped ones)
      row[k] = p;
      if (probs) probs[(((int64_t)h * R + r) * Lq + q) * Lk + k] = p;
    }
  }
  __syncthreads();
  float* vs = ks;  // [Lk][dk]
  for (int idx = tid; idx < Lk * dk; idx += 256) {
    const int k = idx / dk, d = idx % dk;
    vs[k * dk + d] = V[((int64_t)kvr * Lk + k) * ldv + h * dk + d];
  }
  __syncthreads();
  for (int idx = tid; idx < Lq * dk; idx += 256) {
    const int q = idx / dk, d = idx % dk;
    float acc = 0.f;
    const float* prow = ps + q * Lk;
    for (int k = 0; k < Lk; ++k) acc += prow[k] * vs[k * dk + d];
    O[((int64_t)r * Lq + q) * ldo + h * dk + d] = acc;
  }
}

constexpr int ATT_BWD_MAXJ = 32;  // Lk*dk <= 8192

__global__ __launch_bounds__(256) void attention_bwd_kernel(
    const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K, int64_t ldk, const float* __restrict__ V,
    int64_t ldv, const float* __restrict__ dO, int64_t lddo, float* __restrict__ dQ, int64_t lddq,
    float* __restrict__ dK, int64_t lddk, float* __restrict__ dV, int64_t lddv,
    const int64_t* __restrict__ key_tokens, int causal, int R, int Lq, int Lk, int dk, int kv_div, int kv_mod, float p_drop,
    uint32_t salt, const uint64_t* __restrict__ rng_state) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int kvr = blockIdx.x, h = blockIdx.y;
  DropRng rng;
  if (p_drop > 0.f) rng.init(rng_state);
  const int ldd = dk + 1;
  float* qs = sm;                 // [Lq][dk+1]
  float* dos = qs + Lq * ldd;     // [Lq][dk+1]
  float* ks = dos + Lq * ldd;     // [Lk][dk+1]
  float* vs = ks + Lk * ldd;      // [Lk][dk+1]
  float* ps = vs + Lk * ldd;      // [Lq][Lk]
  float* dps = ps + Lq * Lk;      // [Lq][Lk]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < Lk * dk; idx += 256) {
    const int k = idx / dk, d = idx % dk;
    ks[k * ldd + d] = K[((int64_t)kvr * Lk + k) * ldk + h * dk + d];
    vs[k * ldd + d] = V[((int64_t)kvr * Lk + k) * ldv + h * dk + d];
  }
  float dka[ATT_BWD_MAXJ], dva[ATT_BWD_MAXJ];
#pragma unroll
  for (int j = 0; j < ATT_BWD_MAXJ; ++j) { dka[j] = 0.f; dva[j] = 0.f; }
  const float sq = sqrtf((float)dk);
  for (int r = 0; r < R; ++r) {
    if ((r / kv_div) % kv_mod != kvr) continue;
    __syncthreads();
    for (int idx = tid; idx < Lq * dk; idx += 256) {
      const int q = idx / dk, d = idx % dk;
      qs[q * ldd + d] = Q[((int64_t)r * Lq + q) * ldq + h * dk + d];
      dos[q * ldd + d] = dO[((int64_t)r * Lq + q) * lddo + h * dk + d];
    }
    __syncthreads();
    for (int idx = tid; idx < Lq * Lk; idx += 256) {
      const int q = idx / Lk, k = idx % Lk;
      float dot = 0.f, dp = 0.f;
      const float* a = qs + q * ldd;
      const float* b = ks + k * ldd;
      const float* g = dos + q * ldd;
      const float* vv = vs + k * ldd;
      for (int d = 0; d < dk; ++d) { dot += a[d] * b[d]; dp += g[d] * vv[d]; }
      float s = dot / sq;
      const bool masked = (key_tokens && key_tokens[(int64_t)r * Lk + k] == NACF_PAD) || (causal && (k > q || (causal > 1 && k <= q - (causal - 1))));
      if (masked) s = -10e6f;
      ps[idx] = s;
      dps[idx] = dp;
    }
    __syncthreads();
    for (int q = wave; q < Lq; q += 4) {
      float* row = ps + q * Lk;
      float* drow = dps + q * Lk;
      float m = -3.0e38f;
      for (int k = lane; k < Lk; k += 64) m = fmaxf(m, row[k]);
      m = wave_max(m);
      float s = 0.f;
      for (int k = lane; k < Lk; k += 64) { const float e = expf(row[k] - m); row[k] = e; s += e; }
      s = wave_sum(s);
      float dotp = 0.f;
      if (p_drop > 0.f) {
        // O = (P o M) V with M = keep / (1 - p): dP = (dO V^T) o M, the soft-max backward runs on the undropped P, dV on P o M
        const uint64_t e0 = (((uint64_t)r * gridDim.y + h) * (uint64_t)Lq + q) * (uint64_t)Lk;
        for (int k = lane; k < Lk; k += 64) {
          const float p = row[k] / s, mk = rng.keep1(e0 + k, salt, p_drop);
          const float dpk = drow[k] * mk;
          dotp += p * dpk;
          row[k] = p;
          drow[k] = dpk;
        }
        dotp = wave_sum(dotp);
        for (int k = lane; k < Lk; k += 64) {
          const float p = row[k];
          drow[k] = p * (drow[k] - dotp) / sq;
          row[k] = p * rng.keep1(e0 + k, salt, p_drop);
        }
        continue;
      }
      for (int k = lane; k < Lk; k += 64) { const float p = row[k] / s; row[k] = p; dotp += p * drow[k]; }
      dotp = wave_sum(dotp);
      for (int k = lane; k < Lk; k += 64) drow[k] = row[k] * (drow[k] - dotp) / sq;  // grad wrt Q.K^T
    }
    __syncthreads();
    for (int idx = tid; idx < Lq * dk; idx += 256) {
      const int q = idx / dk, d = idx % dk;
      float acc = 0.f;
      const float* ds = dps + q * Lk;
      for (int k = 0; k < Lk; ++k) acc += ds[k] * ks[k * ldd + d];
      dQ[((int64_t)r * Lq + q) * lddq + h * dk + d] = acc;
    }
#pragma unroll
    for (int j = 0; j < ATT_BWD_MAXJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < Lk * dk) {
        const int k = idx / dk, d = idx % dk;
        float a = 0.f, b = 0.f;
        for (int q = 0; q < Lq; ++q) {
          a += dps[q * Lk + k] * qs[q * ldd + d];
          b += ps[q * Lk + k] * dos[q * ldd + d];
        }
        dka[j] += a;
        dva[j] += b;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < ATT_BWD_MAXJ; ++j) {
    const int idx = tid + 256 * j;
    if (idx < Lk * dk) {
      const int k = idx / dk, d = idx % dk;
      dK[((int64_t)kvr * Lk + k) * lddk + h * dk + d] = dka[j];
      dV[((int64_t)kvr * Lk + k) * lddv + h * dk + d] = dva[j];
    }
  }
}

}  // namespace

extern "C" {

int nacf_embed_ln_fwd(const int64_t* tokens, const int64_t* category, const float* additional, const float* word_emb,
                      const float* pos_emb, const float* cat_emb, const float* ln_w, const float* ln_b, float* out,
                      float* xhat, float* rstd, int R, int L, int D, int vdiv, int vmod, float eps, float p_drop,
                      uint32_t salt, const uint64_t* rng_state, nacf_stream_t stream) {
  NACF_CHECK(tokens && word_emb && pos_emb && ln_w && ln_b && out, NACF_EINVAL, "nacf_embed_ln_fwd: null pointer");
  NACF_CHECK(R > 0 && L > 0 && D > 0 && vdiv > 0 && vmod > 0, NACF_EINVAL, "nacf_embed_ln_fwd: bad shape");
  NACF_CHECK(D % 4 == 0 && D <= EMB_MAXJ * EMB_THREADS * 4, NACF_EUNSUPPORTED,
             "nacf_embed_ln_fwd: D must be a multiple of 4 and <= 2048 (got %d)", D);
  NACF_CHECK(p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL, "nacf_embed_ln_fwd: dropout needs rng_state, p<1");
  const bool al = ((reinterpret_cast<uintptr_t>(word_emb) | reinterpret_cast<uintptr_t>(pos_emb) | reinterpret_cast<uintptr_t>(out) |
                    reinterpret_cast<uintptr_t>(ln_w) | reinterpret_cast<uintptr_t>(ln_b)) & 15) == 0;
  const int rows = R * L;
#define NACF_EMB_WAVE(NJ)                                                                                                  \
  hipLaunchKernelGGL(embed_ln_fwd_wave_kernel<NJ>, dim3(cdiv(rows, 4)), dim3(256), 0, as_hip(stream), tokens, category,    \
                     additional, word_emb, pos_emb, cat_emb, ln_w, ln_b, out, xhat, rstd, rows, L, vdiv, vmod, eps, p_drop, \
                     salt, rng_state)
  if (al && D == 256) NACF_EMB_WAVE(1);
  else if (al && D == 512) NACF_EMB_WAVE(2);
  else if (al && D == 768) NACF_EMB_WAVE(3);
  else if (al && D == 1024) NACF_EMB_WAVE(4);
  else
    hipLaunchKernelGGL(embed_ln_fwd_kernel, dim3(R * L), dim3(EMB_THREADS), 0, as_hip(stream), tokens, category,
                       additional, word_emb, pos_emb, cat_emb, ln_w, ln_b, out, xhat, rstd, L, D, vdiv, vmod, eps, p_drop,
                       salt, rng_state);
#undef NACF_EMB_WAVE
  NACF_LAUNCH_CHECK("nacf_embed_ln_fwd");
  return NACF_OK;
}

static int embed_bwd_blocks(int rows) { return rows < 1024 ? rows : 1024; }   // row slabs of the LayerNorm backward (weight/bias gradient partials)

size_t nacf_embed_ln_bwd_workspace(int R, int L, int D) {
  return (size_t)embed_bwd_blocks(R * L) * 2 * D * sizeof(float) + 256;
}

int nacf_embed_ln_bwd(const float* dOut, const float* xhat, const float* rstd, const float* ln_w, float* dE,
                      float* dln_w, float* dln_b, float beta, int R, int L, int D, float p_drop, uint32_t salt,
                      const uint64_t* rng_state, void* ws, size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(dOut && xhat && rstd && ln_w && dE, NACF_EINVAL, "nacf_embed_ln_bwd: null pointer");
  NACF_CHECK(R > 0 && L > 0 && D > 0 && D % 4 == 0 && D <= EMB_MAXJ * EMB_THREADS * 4, NACF_EINVAL,
             "nacf_embed_ln_bwd: bad shape");
  NACF_CHECK(ws && ws_bytes >= nacf_embed_ln_bwd_workspace(R, L, D), NACF_EWORKSPACE, "nacf_embed_ln_bwd: workspace too small");
  NACF_CHECK(p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL, "nacf_embed_ln_bwd: dropout needs rng_state, p<1");
  const int rows = R * L, nblk = embed_bwd_blocks(rows);
  float* part = reinterpret_cast<float*>(ws);
  hipStream_t s = as_hip(stream);
  hipLaunchKernelGGL(embed_ln_bwd_kernel, dim3(nblk), dim3(EMB_THREADS), 0, s, dOut, xhat, rstd, ln_w, dE, part, rows, D,
                     p_drop, salt, rng_state);
  hipLaunchKernelGGL(embed_ln_bwd_final_kernel, dim3(cdiv(D, 16)), dim3(256), 0, s, part, nblk, D, dln_w, dln_b, beta);
  NACF_LAUNCH_CHECK("nacf_embed_ln_bwd");
  return NACF_OK;
}

int nacf_layernorm_fwd(const float* x, const float* ln_w, const float* ln_b, float* out, float* xhat, float* rstd,
                       int rows, int D, float eps, int seg_in, int seg_out, int seg_off, float p_drop, uint32_t salt,
                       const uint64_t* rng_state, const int64_t* row_tokens, nacf_stream_t stream) {
  NACF_CHECK(x && ln_w && ln_b && out && rows > 0 && seg_in > 0 && seg_out >= seg_in && seg_off >= 0, NACF_EINVAL,
             "nacf_layernorm_fwd: bad argument");
  NACF_CHECK(D % 4 == 0 && D > 0 && D <= EMB_MAXJ * EMB_THREADS * 4, NACF_EUNSUPPORTED,
             "nacf_layernorm_fwd: D must be a multiple of 4 and <= 2048 (got %d)", D);
  NACF_CHECK(p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL, "nacf_layernorm_fwd: dropout needs rng_state, p<1");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(rows), dim3(EMB_THREADS), 0, as_hip(stream), x, ln_w, ln_b, out, xhat, rstd, D,
                     eps, seg_in, seg_out, seg_off, p_drop, salt, rng_state, row_tokens);
  NACF_LAUNCH_CHECK("nacf_layernorm_fwd");
  return NACF_OK;
}

size_t nacf_layernorm_bwd_workspace(int rows, int D) {
  return (size_t)embed_bwd_blocks(rows) * 2 * D * sizeof(float) + 256;
}

int nacf_layernorm_bwd(const float* dOut, const float* xhat, const float* rstd, const float* ln_w, float* dX,
                       float* dln_w, float* dln_b, float beta, int rows, int D, int seg_in, int seg_out, int seg_off,
                       float p_drop, uint32_t salt, const uint64_t* rng_state, const int64_t* row_tokens, void* ws,
                       size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(dOut && xhat && rstd && ln_w && dX && rows > 0 && seg_in > 0 && seg_out >= seg_in && seg_off >= 0, NACF_EINVAL,
             "nacf_layernorm_bwd: bad argument");
  NACF_CHECK(D % 4 == 0 && D > 0 && D <= EMB_MAXJ * EMB_THREADS * 4, NACF_EUNSUPPORTED, "nacf_layernorm_bwd: bad D");
  NACF_CHECK(ws && ws_bytes >= nacf_layernorm_bwd_workspace(rows, D), NACF_EWORKSPACE, "nacf_layernorm_bwd: workspace too small");
  NACF_CHECK(p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL, "nacf_layernorm_bwd: dropout needs rng_state, p<1");
  const int nblk = embed_bwd_blocks(rows);
  float* part = reinterpret_cast<float*>(ws);
  hipStream_t s = as_hip(stream);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(EMB_THREADS), 0, s, dOut, xhat, rstd, ln_w, dX, part, rows, D,
                     seg_in, seg_out, seg_off, p_drop, salt, rng_state, row_tokens);
  hipLaunchKernelGGL(embed_ln_bwd_final_kernel, dim3(cdiv(D, 16)), dim3(256), 0, s, part, nblk, D, dln_w, dln_b, beta);
  NACF_LAUNCH_CHECK("nacf_layernorm_bwd");
  return NACF_OK;
}

static int scatter_pos_chunks(int R) { int c = R / 16; return c < 1 ? 1 : (c > 16 ? 16 : c); }

size_t nacf_embed_scatter_bwd_workspace(int R, int L, int D, int n_video) {
  const size_t chunks = (size_t)cdiv(R * L, SCATTER_CHUNK);
  const size_t special = chunks * SPECIAL_N * D;
  const size_t pos = (size_t)scatter_pos_chunks(R) * L * D;
  const size_t vid = (size_t)(n_video > 0 ? n_video : 1) * D;
  return (special + pos + vid) * sizeof(float) + 256;
}

int nacf_embed_scatter_bwd(const float* dE, const int64_t* tokens, const int64_t* category, float* dword, float* dpos,
                           float* dcat, float* dadd, int R, int L, int D, int V, int n_cat, int n_video, int vdiv,
                           int vmod, void* ws, size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(dE && tokens, NACF_EINVAL, "nacf_embed_scatter_bwd: null pointer");
  NACF_CHECK(R > 0 && L > 0 && D > 0 && D % 4 == 0 && D <= EMB_MAXJ * EMB_THREADS * 4 && vdiv > 0 && vmod > 0,
             NACF_EINVAL, "nacf_embed_scatter_bwd: bad shape");
  NACF_CHECK(!(dcat && !category), NACF_EINVAL, "nacf_embed_scatter_bwd: dcat needs category");
  NACF_CHECK(!((dcat || dadd) && n_video <= 0), NACF_EINVAL, "nacf_embed_scatter_bwd: n_video must be > 0");
  NACF_CHECK(ws && ws_bytes >= nacf_embed_scatter_bwd_workspace(R, L, D, n_video), NACF_EWORKSPACE,
             "nacf_embed_scatter_bwd: workspace too small");
  hipStream_t s = as_hip(stream);
  const int rows = R * L;
  const int chunks = cdiv(rows, SCATTER_CHUNK);
  float* part_special = reinterpret_cast<float*>(ws);
  float* part_pos = part_special + (size_t)chunks * SPECIAL_N * D;
  float* vsum_ws = part_pos + (size_t)scatter_pos_chunks(R) * L * D;
  ScatterArgs a = {};
  a.dE = dE; a.tokens = tokens; a.category = category; a.dword = dword; a.dpos = dpos; a.dcat = dcat;
  a.vsum = dadd ? dadd : vsum_ws; a.part_special = part_special; a.part_pos = part_pos;
  a.R = R; a.L = L; a.D = D; a.rows = rows;
  a.chunks = chunks;
  a.n_special = dword ? (V - 1 < SPECIAL_N ? V - 1 : SPECIAL_N) : 0;   // tiny vocabularies
  if (a.n_special < 0) a.n_special = 0;
  a.first_tok = 1 + a.n_special;
  const int pch = scatter_pos_chunks(R);
  a.pos_rows_per = cdiv(R, pch);
  a.pos_real = dpos ? cdiv(R, a.pos_rows_per) : 0;
  a.n_video = (dcat || dadd) ? n_video : 0; a.vdiv = vdiv; a.vmod = vmod; a.n_cat = dcat ? n_cat : 0;
  // phase A: special partials | video sums | position partials | words (8 rows per workgroup)
  a.b_special = 0;
  a.b_video = a.b_special + chunks * a.n_special;
  a.b_pos = a.b_video + a.n_video;
  a.b_word = a.b_pos + L * a.pos_real;
  const int grid_a = a.b_word + (dword ? cdiv(rows, SCAT_THREADS / 64) : 0);
  if (grid_a > 0) hipLaunchKernelGGL(embed_scatter_gather_kernel, dim3(grid_a), dim3(SCAT_THREADS), 0, s, a);
  // phase B: the fixed-order combines
  a.c_special = 0;
  a.c_pos = a.c_special + a.n_special;
  a.c_cat = a.c_pos + (dpos ? L : 0);
  const int grid_b = a.c_cat + a.n_cat;
  if (grid_b > 0) hipLaunchKernelGGL(embed_scatter_combine_kernel, dim3(grid_b), dim3(EMB_THREADS), 0, s, a);
  NACF_LAUNCH_CHECK("nacf_embed_scatter_bwd");
  return NACF_OK;
}

int nacf_masked_mean_fwd(const float* y, const int64_t* tokens, float* out, int R, int L, int D, nacf_stream_t stream) {
  NACF_CHECK(y && tokens && out && R > 0 && L > 0 && D > 0, NACF_EINVAL, "nacf_masked_mean_fwd: bad argument");
  hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3(R), dim3(256), 0, as_hip(stream), y, tokens, out, L, D);
  NACF_LAUNCH_CHECK("nacf_masked_mean_fwd");
  return NACF_OK;
}

int nacf_attention_fwd_dropout(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O,
                               int64_t ldo, const int64_t* key_tokens, int causal, float* probs, int R, int H, int Lq, int Lk,
                               int dk, int kv_div, int kv_mod, float p_drop, uint32_t salt, const uint64_t* rng_state,
                               nacf_stream_t stream) {
  NACF_CHECK(Q && K && V && O, NACF_EINVAL, "nacf_attention_fwd: null pointer");
  NACF_CHECK(p_drop >= 0.f && p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL,
             "nacf_attention_fwd: dropout needs 0 <= p < 1 and an rng state");
  NACF_CHECK(R > 0 && H > 0 && Lq > 0 && Lk > 0 && dk > 0 && kv_div > 0 && kv_mod > 0, NACF_EINVAL,
             "nacf_attention_fwd: bad shape");
  NACF_CHECK(!(causal && Lq != Lk), NACF_EINVAL, "nacf_attention_fwd: causal mask needs Lq == Lk");
  // more than 32 queries per sequence run as blocks of 32 (one wave each) when no causal mask ties a query to its index
  // (probability dropout runs on the LDS-tile kernel below: the probabilities sit in LDS there, one mask multiply away)
  if (p_drop == 0.f && attn_mfma_ok(causal ? Lq : min(Lq, 32), Lk, dk) && attn_aligned(Q, ldq) && attn_aligned(K, ldk) &&
      attn_aligned(V, ldv) && attn_aligned(O, ldo)) {
    {
      // cross-attention over a long memory: one workgroup per (memory row set, head) stages K / V in LDS once and its
      // waves serve every sequence that attends to it (attn_mfma.hpp:fwd_lds_kernel); NACF_ATTN_LDS=0 keeps fwd_kernel
      const char* e = getenv("NACF_ATTN_LDS");
      const int groups = cdiv(R, kv_div);
      const int n_kv = kv_mod < groups ? kv_mod : groups;
      const int nseq_max = kv_div * cdiv(groups, kv_mod);
      // (worth it when the row set feeds all four waves: decode, 4 query blocks per video -- 80 -> 63 us; with the two
      //  sequences per video of a training step half the waves idle and fwd_kernel's 39 us becomes 52 us)
      if (Lk > 32 && !key_tokens && !causal && !probs && dk == 64 && nseq_max * cdiv(Lq, 32) >= 4 && !(e && atoi(e) == 0)) {
        const size_t lds = (size_t)2 * 128 * attn::FL_PITCH * sizeof(float);
        static bool set_fl = false;
        if (!set_fl) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn::fwd_lds_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn::fwd_lds_kernel<4, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          set_fl = true;
        }
        if (attn_precision(dk))
          hipLaunchKernelGGL((attn::fwd_lds_kernel<4, 1>), dim3(n_kv * H), dim3(256), lds, as_hip(stream), Q, ldq, K, ldk, V, ldv,
                             O, ldo, R, n_kv, H, Lq, Lk, kv_div, kv_mod, nseq_max);
        else
          hipLaunchKernelGGL((attn::fwd_lds_kernel<4>), dim3(n_kv * H), dim3(256), lds, as_hip(stream), Q, ldq, K, ldk, V, ldv,
                             O, ldo, R, n_kv, H, Lq, Lk, kv_div, kv_mod, nseq_max);
        NACF_LAUNCH_CHECK("nacf_attention_fwd(mfma, lds)");
        return NACF_OK;
      }
    }
    // matrix-core path: one wave per (sequence, head, 32 queries), operands straight from HBM/L2 into MFMA fragments
    const int nqb = cdiv(Lq, 32);
    const dim3 grid(cdiv(R * H * nqb, 4));
    hipStream_t s = as_hip(stream);
#define NACF_ATTN_FWD(NKT, DK16, PR)                                                                                      \
  hipLaunchKernelGGL((attn::fwd_kernel<NKT, DK16, PR>), grid, dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, key_tokens, \
                     causal, probs, R, H, Lq, Lk, kv_div, kv_mod, nqb)
    const int pr = attn_precision(dk);
    if (Lk <= 32) { if (dk == 64) { if (pr) NACF_ATTN_FWD(2, 4, 1); else NACF_ATTN_FWD(2, 4, 0); } else NACF_ATTN_FWD(2, 1, 0); }
    else { if (dk == 64) { if (pr) NACF_ATTN_FWD(8, 4, 1); else NACF_ATTN_FWD(8, 4, 0); } else NACF_ATTN_FWD(8, 1, 0); }
#undef NACF_ATTN_FWD
    NACF_LAUNCH_CHECK("nacf_attention_fwd(mfma)");
    return NACF_OK;
  }
  const size_t lds = ((size_t)Lq * (dk + 1) + (size_t)Lk * (dk + 1) + (size_t)Lq * Lk) * sizeof(float);
  NACF_CHECK(lds <= 160 * 1024, NACF_EUNSUPPORTED, "nacf_attention_fwd: tile does not fit LDS (%zu B)", lds);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(R, H), dim3(256), lds, as_hip(stream), Q, ldq, K, ldk, V, ldv, O, ldo,
                     key_tokens, causal, probs, R, Lq, Lk, dk, kv_div, kv_mod, p_drop, salt, rng_state);
  NACF_LAUNCH_CHECK("nacf_attention_fwd");
  return NACF_OK;
}

int nacf_attention_fwd(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O,
                       int64_t ldo, const int64_t* key_tokens, int causal, float* probs, int R, int H, int Lq, int Lk,
                       int dk, int kv_div, int kv_mod, nacf_stream_t stream) {
  return nacf_attention_fwd_dropout(Q, ldq, K, ldk, V, ldv, O, ldo, key_tokens, causal, probs, R, H, Lq, Lk, dk, kv_div, kv_mod,
                                    0.f, 0u, nullptr, stream);
}

int nacf_attention_bwd_dropout(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv,
                               const float* dO, int64_t lddo, float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV,
                               int64_t lddv, const int64_t* key_tokens, int causal, int R, int n_kv, int H, int Lq, int Lk,
                               int dk, int kv_div, int kv_mod, float p_drop, uint32_t salt, const uint64_t* rng_state,
                               nacf_stream_t stream) {
  NACF_CHECK(Q && K && V && dO && dQ && dK && dV, NACF_EINVAL, "nacf_attention_bwd: null pointer");
  NACF_CHECK(p_drop >= 0.f && p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL,
             "nacf_attention_bwd: dropout needs 0 <= p < 1 and an rng state");
  NACF_CHECK(R > 0 && n_kv > 0 && H > 0 && Lq > 0 && Lk > 0 && dk > 0 && kv_div > 0 && kv_mod > 0, NACF_EINVAL,
             "nacf_attention_bwd: bad shape");
  NACF_CHECK(n_kv == kv_mod || (kv_div == 1 && kv_mod >= R && n_kv == R), NACF_EINVAL,
             "nacf_attention_bwd: n_kv must equal kv_mod (or R for the identity map)");
  NACF_CHECK(Lk * dk <= ATT_BWD_MAXJ * 256, NACF_EUNSUPPORTED, "nacf_attention_bwd: Lk*dk > %d", ATT_BWD_MAXJ * 256);
  NACF_CHECK(!(causal && Lq != Lk), NACF_EINVAL, "nacf_attention_bwd: causal mask needs Lq == Lk");
  if (p_drop == 0.f && attn_mfma_ok(Lq, Lk, dk) && attn_aligned(Q, ldq) && attn_aligned(K, ldk) && attn_aligned(V, ldv) &&
      attn_aligned(dO, lddo) && attn_aligned(dQ, lddq) && attn_aligned(dK, lddk) && attn_aligned(dV, lddv)) {
    // sequences per memory row set (upper bound) -> waves per item (1, 2 or 4) and rounds of the item loop
    const int groups = cdiv(R, kv_div);
    const int nseq_max = kv_div * cdiv(groups, kv_mod);
    {
      // cross-attention over a long memory (no key mask, not causal): the four waves of a workgroup split the keys of one
      // (memory row set, head) -- attn_mfma.hpp:bwd_kb_kernel (NACF train step: 145 -> 97 us); NACF_ATTN_KB=0 keeps the
      // one-wave-per-item kernel.
      const char* e = getenv("NACF_ATTN_KB");
      if (Lk > 32 && !key_tokens && !causal && dk == 64 && !(e && atoi(e) == 0)) {
        const size_t lds_kb = (size_t)(3 * 4 * 32 + 4 * 32 * attn::KB_RED_PITCH) * sizeof(float);   // tiles alias the partials
        static bool set_kb = false;
        if (!set_kb) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn::bwd_kb_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn::bwd_kb_kernel<4, 0>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn::bwd_kb_kernel<4, 1, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          set_kb = true;
        }
        if (e && atoi(e) == 2)
          hipLaunchKernelGGL((attn::bwd_kb_kernel<4, 0>), dim3(n_kv * H), dim3(256), lds_kb, as_hip(stream), Q, ldq, K, ldk, V,
                             ldv, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, R, n_kv, H, Lq, Lk, kv_div, kv_mod, nseq_max);
        else if (attn_precision(dk))
          hipLaunchKernelGGL((attn::bwd_kb_kernel<4, 1, 1>), dim3(n_kv * H), dim3(256), lds_kb, as_hip(stream), Q, ldq, K, ldk, V,
                             ldv, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, R, n_kv, H, Lq, Lk, kv_div, kv_mod, nseq_max);
        else
          hipLaunchKernelGGL((attn::bwd_kb_kernel<4>), dim3(n_kv * H), dim3(256), lds_kb, as_hip(stream), Q, ldq, K, ldk, V,
                             ldv, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, R, n_kv, H, Lq, Lk, kv_div, kv_mod, nseq_max);
        NACF_LAUNCH_CHECK("nacf_attention_bwd(mfma, key blocks)");
        return NACF_OK;
      }
    }
    // measured on MI355X (NACF train step, 2 sequences per video): 1 wave per item 144 us, 2 waves 187 us -- the ordered
    // dK / dV turns and their barriers cost more than the extra waves hide; NACF_ATTN_WPI=2|4 keeps the variant testable
    int wpi = 1;
    { const char* e = getenv("NACF_ATTN_WPI"); if (e && (atoi(e) == 2 || atoi(e) == 4) && nseq_max >= atoi(e)) wpi = atoi(e); }
    const int rounds = cdiv(nseq_max, wpi);
    const dim3 grid(cdiv(n_kv * H * wpi, 4));
    hipStream_t s = as_hip(stream);
    const int nkt = Lk <= 32 ? 2 : 8;
    const size_t lds_m = (size_t)4 * 32 * (nkt * 16 + 16) * sizeof(float);   // one transpose tile per wave
#define NACF_ATTN_BWD(NKT, DK16, PR)                                                                                   \
  do {                                                                                                                \
    static bool set_##NKT##_##DK16##_##PR = false;                                                                    \
    if (!set_##NKT##_##DK16##_##PR) {                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn::bwd_kernel<NKT, DK16, PR>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
      set_##NKT##_##DK16##_##PR = true;                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((attn::bwd_kernel<NKT, DK16, PR>), grid, dim3(256), lds_m, s, Q, ldq, K, ldk, V, ldv, dO, lddo, \
                       dQ, lddq, dK, lddk, dV, lddv, key_tokens, causal, R, n_kv, H, Lq, Lk, kv_div, kv_mod, wpi,      \
                       rounds);                                                                                       \
  } while (0)
    const int pr = attn_precision(dk);
    if (nkt == 2) { if (dk == 64) { if (pr) NACF_ATTN_BWD(2, 4, 1); else NACF_ATTN_BWD(2, 4, 0); } else NACF_ATTN_BWD(2, 1, 0); }
    else { if (dk == 64) { if (pr) NACF_ATTN_BWD(8, 4, 1); else NACF_ATTN_BWD(8, 4, 0); } else NACF_ATTN_BWD(8, 1, 0); }
#undef NACF_ATTN_BWD
    NACF_LAUNCH_CHECK("nacf_attention_bwd(mfma)");
    return NACF_OK;
  }
  const size_t lds = ((size_t)2 * Lq * (dk + 1) + (size_t)2 * Lk * (dk + 1) + (size_t)2 * Lq * Lk) * sizeof(float);
  NACF_CHECK(lds <= 160 * 1024, NACF_EUNSUPPORTED, "nacf_attention_bwd: tile does not fit LDS (%zu B)", lds);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(n_kv, H), dim3(256), lds, as_hip(stream), Q, ldq, K, ldk, V, ldv, dO, lddo,
                     dQ, lddq, dK, lddk, dV, lddv, key_tokens, causal, R, Lq, Lk, dk, kv_div, kv_mod, p_drop, salt, rng_state);
  NACF_LAUNCH_CHECK("nacf_attention_bwd");
  return NACF_OK;
}

int nacf_attention_bwd(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv,
                       const float* dO, int64_t lddo, float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV,
                       int64_t lddv, const int64_t* key_tokens, int causal, int R, int n_kv, int H, int Lq, int Lk,
                       int dk, int kv_div, int kv_mod, nacf_stream_t stream) {
  return nacf_attention_bwd_dropout(Q, ldq, K, ldk, V, ldv, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, key_tokens, causal, R, n_kv, H,
                                    Lq, Lk, dk, kv_div, kv_mod, 0.f, 0u, nullptr, stream);
}

}  // extern "C"

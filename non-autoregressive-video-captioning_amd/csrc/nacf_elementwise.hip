// HBM-bound kernels of the NACF path: HighWay mix, BatchNorm + temporal concat,
// time mean, narrow log-softmax / KL, epilogue backward, wide (vocabulary)
// log-softmax + NLL + cross-entropy backward, fused clip + Adam.
// All reductions use a fixed summation order (no float atomics).
#include "common.hpp"

namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ------------------------------------------------------------------ HighWay
__global__ void highway_mix_fwd_kernel(const float* __restrict__ H, const float* __restrict__ TG,
                                       float* __restrict__ out, int rows, int D, float p, uint32_t salt,
                                       const uint64_t* __restrict__ rng_state) {
  const int64_t total = (int64_t)rows * D;
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / D), d = (int)(e % D);
    const float h = H[e];
    const float t = TG[(int64_t)r * 2 * D + d];
    const float g = TG[(int64_t)r * 2 * D + D + d];
    float o = g * h + (1.f - g) * t;
    if (p > 0.f) o *= rng.keep1((uint64_t)e, salt, p);
    out[e] = o;
  }
}

__global__ void highway_mix_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ H,
                                       const float* __restrict__ TG, float* __restrict__ dH,
                                       float* __restrict__ dP, int rows, int D, float p, uint32_t salt,
                                       const uint64_t* __restrict__ rng_state) {
  const int64_t total = (int64_t)rows * D;
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / D), d = (int)(e % D);
    float g0 = dOut[e];
    if (p > 0.f) g0 *= rng.keep1((uint64_t)e, salt, p);
    const float h = H[e];
    const float t = TG[(int64_t)r * 2 * D + d];
    const float g = TG[(int64_t)r * 2 * D + D + d];
    dH[e] = g0 * g;
    dP[(int64_t)r * 2 * D + d] = g0 * (1.f - g) * (1.f - t * t);
    dP[(int64_t)r * 2 * D + D + d] = g0 * (h - t) * g * (1.f - g);
  }
}

// float4 variants (D % 4 == 0, 16-byte aligned buffers): one Philox call per 4 elements instead of per element
__global__ void highway_mix_fwd_v4_kernel(const float* __restrict__ H, const float* __restrict__ TG,
                                          float* __restrict__ out, int rows, int D, float p, uint32_t salt,
                                          const uint64_t* __restrict__ rng_state) {
  const int D4 = D >> 2;
  const int64_t total = (int64_t)rows * D4;
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / D4), d = (int)(e % D4) * 4;
    const f32x4 h = *reinterpret_cast<const f32x4*>(H + (int64_t)r * D + d);
    const f32x4 t = *reinterpret_cast<const f32x4*>(TG + (int64_t)r * 2 * D + d);
    const f32x4 g = *reinterpret_cast<const f32x4*>(TG + (int64_t)r * 2 * D + D + d);
    f32x4 o = g * h + (1.f - g) * t;
    if (p > 0.f) o *= rng.keep4((uint64_t)e, salt, p);
    *reinterpret_cast<f32x4*>(out + (int64_t)r * D + d) = o;
  }
}
__global__ void highway_mix_bwd_v4_kernel(const float* __restrict__ dOut, const float* __restrict__ H,
                                          const float* __restrict__ TG, float* __restrict__ dH,
                                          float* __restrict__ dP, int rows, int D, float p, uint32_t salt,
                                          const uint64_t* __restrict__ rng_state) {
  const int D4 = D >> 2;
  const int64_t total = (int64_t)rows * D4;
  DropRng rng;
  if (p > 0.f) rng.init(rng_state);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / D4), d = (int)(e % D4) * 4;
    f32x4 g0 = *reinterpret_cast<const f32x4*>(dOut + (int64_t)r * D + d);
    if (p > 0.f) g0 *= rng.keep4((uint64_t)e, salt, p);
    const f32x4 h = *reinterpret_cast<const f32x4*>(H + (int64_t)r * D + d);
    const f32x4 t = *reinterpret_cast<const f32x4*>(TG + (int64_t)r * 2 * D + d);
    const f32x4 g = *reinterpret_cast<const f32x4*>(TG + (int64_t)r * 2 * D + D + d);
    *reinterpret_cast<f32x4*>(dH + (int64_t)r * D + d) = g0 * g;
    *reinterpret_cast<f32x4*>(dP + (int64_t)r * 2 * D + d) = g0 * (1.f - g) * (1.f - t * t);
    *reinterpret_cast<f32x4*>(dP + (int64_t)r * 2 * D + D + d) = g0 * (h - t) * g * (1.f - g);
  }
}

// ------------------------------------------------------------------ BatchNorm + concat
// stage A: part_sum[s][d] over a slab of rows
__global__ void bn_partial_sum_kernel(const float* __restrict__ x, int rows, int D, int rows_per,
                                      float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float acc = 0.f;
  if (c < D)
    for (int r = r0 + rl; r < r1; r += 4) acc += x[(int64_t)r * D + c];
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < D)
    part[(int64_t)blockIdx.y * D + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// stage B: mean from partials, then partial sum of squared deviations
__global__ void bn_partial_sqdev_kernel(const float* __restrict__ x, int rows, int D, int rows_per, int S,
                                        const float* __restrict__ part_sum, float* __restrict__ part_sq, float n_stat) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float mean = 0.f;
  if (c < D) {
    for (int s = 0; s < S; ++s) mean += part_sum[(int64_t)s * D + c];
    mean /= n_stat;
  }
  float acc = 0.f;
  if (c < D)
    for (int r = r0 + rl; r < r1; r += 4) {
      const float dv = x[(int64_t)r * D + c] - mean;
      acc += dv * dv;
    }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < D)
    part_sq[(int64_t)blockIdx.y * D + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// stage C: finalise statistics, normalise, write into the concatenated memory
__global__ void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int F, int D,
                                int M_total, int f_off, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                int64_t* __restrict__ nbt, float* __restrict__ save_mean,
                                float* __restrict__ save_invstd, int training, float momentum, float eps, int S,
                                const float* __restrict__ part_sum, const float* __restrict__ part_sq, int rows_per,
                                float n_stat) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int rows = B * F;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  if (c >= D) return;
  float mean, invstd;
  if (training) {
    float sm = 0.f, sq = 0.f;
    for (int s = 0; s < S; ++s) { sm += part_sum[(int64_t)s * D + c]; sq += part_sq[(int64_t)s * D + c]; }
    mean = sm / n_stat;
    const float var_b = sq / n_stat;
    invstd = 1.0f / sqrtf(var_b + eps);
    if (blockIdx.y == 0 && rl == 0) {
      const float var_u = sq / (n_stat > 1.f ? n_stat - 1.f : 1.f);
      if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_u;
      if (save_mean) save_mean[c] = mean;
      if (save_invstd) save_invstd[c] = invstd;
      if (nbt && c == 0) nbt[0] += 1;
    }
  } else {
    mean = running_mean[c];
    invstd = 1.0f / sqrtf(running_var[c] + eps);
  }
  const float ww = w ? w[c] : 1.f, bb = b ? b[c] : 0.f;
  for (int r = r0 + rl; r < r1; r += 4) {
    const int bi = r / F, f = r % F;
    out[((int64_t)bi * M_total + f_off + f) * D + c] = (x[(int64_t)r * D + c] - mean) * invstd * ww + bb;
  }
}

// backward stage A: partial sums of dy and dy*xhat
__global__ void bn_bwd_partial_kernel(const float* __restrict__ dOut, const float* __restrict__ x, int B, int F,
                                      int D, int M_total, int f_off, const float* __restrict__ save_mean,
                                      const float* __restrict__ save_invstd, int rows_per,
                                      float* __restrict__ part_dy, float* __restrict__ part_dyx) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int rows = B * F;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float a0 = 0.f, a1 = 0.f;
  if (c < D) {
    const float mean = save_mean[c], invstd = save_invstd[c];
    for (int r = r0 + rl; r < r1; r += 4) {
      const int bi = r / F, f = r % F;
      const float dy = dOut[((int64_t)bi * M_total + f_off + f) * D + c];
      a0 += dy;
      a1 += dy * (x[(int64_t)r * D + c] - mean) * invstd;
    }
  }
  red[0][rl][threadIdx.x & 63] = a0;
  red[1][rl][threadIdx.x & 63] = a1;
  __syncthreads();
  if (rl == 0 && c < D) {
    const int t = threadIdx.x;
    part_dy[(int64_t)blockIdx.y * D + c] = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
    part_dyx[(int64_t)blockIdx.y * D + c] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
  }
}
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ x,
                                    float* __restrict__ dx, int B, int F, int D, int M_total, int f_off,
                                    const float* __restrict__ w, const float* __restrict__ save_mean,
                                    const float* __restrict__ save_invstd, float* __restrict__ dweight,
                                    float* __restrict__ dbias, float beta, int S, int rows_per,
                                    const float* __restrict__ part_dy, const float* __restrict__ part_dyx, float n_stat) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int rows = B * F;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  if (c >= D) return;
  float sdy = 0.f, sdyx = 0.f;
  for (int s = 0; s < S; ++s) { sdy += part_dy[(int64_t)s * D + c]; sdyx += part_dyx[(int64_t)s * D + c]; }
  if (blockIdx.y == 0 && rl == 0) {
    if (dweight) dweight[c] = (beta != 0.f) ? sdyx + beta * dweight[c] : sdyx;
    if (dbias) dbias[c] = (beta != 0.f) ? sdy + beta * dbias[c] : sdy;
  }
  const float mean = save_mean[c], invstd = save_invstd[c];
  const float ww = w ? w[c] : 1.f;
  const float invn = 1.f / n_stat;
  for (int r = r0 + rl; r < r1; r += 4) {
    const int bi = r / F, f = r % F;
    const float dy = dOut[((int64_t)bi * M_total + f_off + f) * D + c];
    const float xh = (x[(int64_t)r * D + c] - mean) * invstd;
    dx[(int64_t)r * D + c] = ww * invstd * (dy - sdy * invn - xh * sdyx * invn);
  }
}

// ---- float4 variants of the five BatchNorm kernels (D % 4 == 0, 16-byte aligned buffers): a workgroup is 16 float4
// columns x 16 row groups (group rg walks rows r0+rg, r0+rg+16, ...), partials folded through LDS in fixed order.
// Same two-pass statistics as above; ~4x fewer, 4x wider loads per thread.
__device__ __forceinline__ f32x4 bn_fold16(f32x4 v, f32x4 (*red)[16], int rg, int c4) {
  red[rg][c4] = v;
  __syncthreads();
  f32x4 t = red[0][c4];
#pragma unroll
  for (int i = 1; i < 16; ++i) t += red[i][c4];
  __syncthreads();
  return t;   // every thread gets the column's total
}
// column totals of the S slab partials: row group rg adds slabs rg, rg+16, ..., the 16 groups fold through LDS
// (a 32-long chain of dependent loads at the start of every workgroup cost more than the workgroup's own rows)
__device__ __forceinline__ f32x4 bn_sum_parts(const float* __restrict__ part, int S, int D, int c, bool valid,
                                              f32x4 (*red)[16], int rg, int c4) {
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  if (valid)
    for (int s = rg; s < S; s += 16) t += *reinterpret_cast<const f32x4*>(part + (int64_t)s * D + c);
  return bn_fold16(t, red, rg, c4);
}
// One launch serves every modality of a joint representation (blockIdx.z): each of these kernels is a chain of ~15
// dependent row loads per thread, bound by latency, not by the 2 x 16 MB it moves -- two modalities side by side cost
// what one does.
constexpr int BN_MAX_MODS = 4;
struct BnMod {
  const float* x; float* dx;
  int F, f_off, rows_per, S;      // S: row slabs this launch walks (grid.y)
  int SP; float n_stat;           // slabs of partials to fold (1: an already folded / all-reduced vector), rows behind them
  const float* w; const float* b;
  float* running_mean; float* running_var; int64_t* nbt;
  float* save_mean; float* save_invstd;
  float* dweight; float* dbias;
  const float* part_in0; const float* part_in1; float* part_out0; float* part_out1;
};
struct BnMods { BnMod m[BN_MAX_MODS]; };

__global__ __launch_bounds__(256) void bn_partial_sum_v4_kernel(BnMods t, int B, int D) {
  __shared__ f32x4 red[16][16];
  const BnMod& m = t.m[blockIdx.z];
  if ((int)blockIdx.y >= m.S) return;
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  const int rows = B * m.F;
  const int r0 = blockIdx.y * m.rows_per, r1 = min(rows, r0 + m.rows_per);
  const float* __restrict__ x = m.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < D) {
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) acc += *reinterpret_cast<const f32x4*>(x + (int64_t)r * D + c);
  }
  const f32x4 tt = bn_fold16(acc, red, rg, c4);
  if (rg == 0 && c < D) *reinterpret_cast<f32x4*>(m.part_out0 + (int64_t)blockIdx.y * D + c) = tt;
}
__global__ __launch_bounds__(256) void bn_partial_sqdev_v4_kernel(BnMods t, int B, int D) {
  __shared__ f32x4 red[16][16];
  const BnMod& m = t.m[blockIdx.z];
  if ((int)blockIdx.y >= m.S) return;
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  const int rows = B * m.F;
  const int r0 = blockIdx.y * m.rows_per, r1 = min(rows, r0 + m.rows_per);
  const float* __restrict__ x = m.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 mean = bn_sum_parts(m.part_in0, m.SP, D, c, c < D, red, rg, c4) / m.n_stat;
  if (c < D) {
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(x + (int64_t)r * D + c) - mean;
      acc += dv * dv;
    }
  }
  const f32x4 tt = bn_fold16(acc, red, rg, c4);
  if (rg == 0 && c < D) *reinterpret_cast<f32x4*>(m.part_out1 + (int64_t)blockIdx.y * D + c) = tt;
}
__global__ __launch_bounds__(256) void bn_apply_v4_kernel(BnMods t, float* __restrict__ out, int B, int D, int M_total,
                                                          int training, float momentum, float eps) {
  const BnMod& m = t.m[blockIdx.z];
  if ((int)blockIdx.y >= m.S) return;
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  __shared__ f32x4 red[16][16];
  const int F = m.F, f_off = m.f_off;
  const int rows = B * F;
  const float n_stat = m.n_stat;
  const int r0 = blockIdx.y * m.rows_per, r1 = min(rows, r0 + m.rows_per);
  const float* __restrict__ x = m.x;
  f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
  if (training) {            // block-uniform: every thread takes part in the folds
    sm = bn_sum_parts(m.part_in0, m.SP, D, c, c < D, red, rg, c4);
    sq = bn_sum_parts(m.part_in1, m.SP, D, c, c < D, red, rg, c4);
  }
  if (c >= D) return;
  f32x4 mean, invstd;
  if (training) {
    mean = sm / n_stat;
    const f32x4 var_b = sq / n_stat;
#pragma unroll
    for (int e = 0; e < 4; ++e) invstd[e] = 1.0f / sqrtf(var_b[e] + eps);
    if (blockIdx.y == 0 && rg == 0) {
      const f32x4 var_u = sq / (n_stat > 1.f ? n_stat - 1.f : 1.f);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (m.running_mean) m.running_mean[c + e] = (1.f - momentum) * m.running_mean[c + e] + momentum * mean[e];
        if (m.running_var) m.running_var[c + e] = (1.f - momentum) * m.running_var[c + e] + momentum * var_u[e];
        if (m.save_mean) m.save_mean[c + e] = mean[e];
        if (m.save_invstd) m.save_invstd[c + e] = invstd[e];
      }
      if (m.nbt && c == 0) m.nbt[0] += 1;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { mean[e] = m.running_mean[c + e]; invstd[e] = 1.0f / sqrtf(m.running_var[c + e] + eps); }
  }
  f32x4 ww = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
  if (m.w) ww = *reinterpret_cast<const f32x4*>(m.w + c);
  if (m.b) bb = *reinterpret_cast<const f32x4*>(m.b + c);
#pragma unroll 4
  for (int r = r0 + rg; r < r1; r += 16) {
    const int bi = r / F, f = r % F;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)r * D + c);
    *reinterpret_cast<f32x4*>(out + ((int64_t)bi * M_total + f_off + f) * D + c) = (v - mean) * invstd * ww + bb;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_partial_v4_kernel(BnMods t, const float* __restrict__ dOut, int B, int D, int M_total) {
  __shared__ f32x4 red[16][16];
  const BnMod& m = t.m[blockIdx.z];
  if ((int)blockIdx.y >= m.S) return;
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  const int F = m.F, f_off = m.f_off;
  const int rows = B * F;
  const int r0 = blockIdx.y * m.rows_per, r1 = min(rows, r0 + m.rows_per);
  const float* __restrict__ x = m.x;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (c < D) {
    const f32x4 mean = *reinterpret_cast<const f32x4*>(m.save_mean + c), invstd = *reinterpret_cast<const f32x4*>(m.save_invstd + c);
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
      const int bi = r / F, f = r % F;
      const f32x4 dy = *reinterpret_cast<const f32x4*>(dOut + ((int64_t)bi * M_total + f_off + f) * D + c);
      a0 += dy;
      a1 += dy * (*reinterpret_cast<const f32x4*>(x + (int64_t)r * D + c) - mean) * invstd;
    }
  }
  const f32x4 t0 = bn_fold16(a0, red, rg, c4);
  const f32x4 t1 = bn_fold16(a1, red, rg, c4);
  if (rg == 0 && c < D) {
    *reinterpret_cast<f32x4*>(m.part_out0 + (int64_t)blockIdx.y * D + c) = t0;
    *reinterpret_cast<f32x4*>(m.part_out1 + (int64_t)blockIdx.y * D + c) = t1;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(BnMods t, const float* __restrict__ dOut, int B, int D, int M_total,
                                                              float beta) {
  const BnMod& m = t.m[blockIdx.z];
  if ((int)blockIdx.y >= m.S) return;
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  const int F = m.F, f_off = m.f_off;
  const int rows = B * F;
  const float n_stat = m.n_stat;
  const int r0 = blockIdx.y * m.rows_per, r1 = min(rows, r0 + m.rows_per);
  const float* __restrict__ x = m.x;
  float* __restrict__ dx = m.dx;
  __shared__ f32x4 red[16][16];
  const f32x4 sdy = bn_sum_parts(m.part_in0, m.SP, D, c, c < D, red, rg, c4);
  const f32x4 sdyx = bn_sum_parts(m.part_in1, m.SP, D, c, c < D, red, rg, c4);
  if (c >= D) return;
  if (blockIdx.y == 0 && rg == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (m.dweight) m.dweight[c + e] = (beta != 0.f) ? sdyx[e] + beta * m.dweight[c + e] : sdyx[e];
      if (m.dbias) m.dbias[c + e] = (beta != 0.f) ? sdy[e] + beta * m.dbias[c + e] : sdy[e];
    }
  }
  const f32x4 mean = *reinterpret_cast<const f32x4*>(m.save_mean + c), invstd = *reinterpret_cast<const f32x4*>(m.save_invstd + c);
  f32x4 ww = {1.f, 1.f, 1.f, 1.f};
  if (m.w) ww = *reinterpret_cast<const f32x4*>(m.w + c);
  const float invn = 1.f / n_stat;
#pragma unroll 4
  for (int r = r0 + rg; r < r1; r += 16) {
    const int bi = r / F, f = r % F;
    const f32x4 dy = *reinterpret_cast<const f32x4*>(dOut + ((int64_t)bi * M_total + f_off + f) * D + c);
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + (int64_t)r * D + c) - mean) * invstd;
    *reinterpret_cast<f32x4*>(dx + (int64_t)r * D + c) = ww * invstd * (dy - sdy * invn - xh * sdyx * invn);
  }
}

// column totals of S slab partials in fixed order: out[d] = sum_s part[s][d]; optionally acc[d] = beta * acc[d] + total
// (data-parallel BatchNorm: the vectors the ranks exchange, and the LOCAL weight / bias gradients)
__global__ void bn_fold_parts_kernel(const float* __restrict__ part, int S, int D, float* __restrict__ out,
                                     float* __restrict__ acc, float beta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += part[(int64_t)s * D + c];
  if (out) out[c] = t;
  if (acc) acc[c] = (beta != 0.f) ? t + beta * acc[c] : t;
}

// the same for every (modality, statistic) of a joint representation in one launch: grid (cdiv(D, 256), n_mod, 2)
struct BnFold {
  const float* part[BN_MAX_MODS][2];
  int S[BN_MAX_MODS];
  float* out[BN_MAX_MODS][2];
  float* acc[BN_MAX_MODS][2];
};
__global__ void bn_fold_multi_kernel(BnFold t, int D, float beta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  const int mod = blockIdx.y, st = blockIdx.z;
  const float* __restrict__ part = t.part[mod][st];
  float v = 0.f;
  for (int s = 0; s < t.S[mod]; ++s) v += part[(int64_t)s * D + c];
  if (t.out[mod][st]) t.out[mod][st][c] = v;
  float* acc = t.acc[mod][st];
  if (acc) acc[c] = (beta != 0.f) ? v + beta * acc[c] : v;
}

// data-parallel BatchNorm, forward statistics from ONE exchange: every rank contributes (sum, squared deviations about its
// OWN mean) of its rows; the parallel-variance merge (Chan et al.) is exact about the global mean:
//   S = sum_i s_i,  mean = S / N,  Q = sum_i [ q_i + n_i (s_i / n_i - mean)^2 ]        (ranks in fixed order)
// gathered: [world][rank_stride] with rank r's (S_r | Q_r) = [2][n_mod][D] first; out: [2][n_mod][D] = (S | Q), what
// nacf_bn_concat_fwd_sync consumes.  rank_stride >= 2 n_mod D + n_mod: the n_mod floats behind a rank's statistics are ITS row
// counts, which travelled in the same all-gather -- every rank must hold rows.n[mod] rows (the merge and the normalisation use
// n x world).  A rank that does not (a ragged global batch) turns the merged statistics into NaN -- the step's loss is NaN on
// every rank, at once, with no extra collective and no host read -- and raises *flag (read by the host at its next sync point).
struct BnMergeN { float n[BN_MAX_MODS]; };
__global__ void bn_sync_merge_kernel(const float* __restrict__ gathered, int world, int n_mod, int D, BnMergeN rows, int64_t rank_stride,
                                     float* __restrict__ out, int* __restrict__ flag) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int C = n_mod * D;
  if (c >= C) return;
  const int mod = c / D;
  const float n = rows.n[mod];
  bool ragged = false;
  if (rank_stride >= 2 * (int64_t)C + n_mod)
    for (int r = 0; r < world; ++r) ragged = ragged || gathered[(int64_t)r * rank_stride + 2 * C + mod] != n;
  float S = 0.f;
  for (int r = 0; r < world; ++r) S += gathered[(int64_t)r * rank_stride + c];
  const float mean = S / (n * (float)world);
  float Q = 0.f;
  for (int r = 0; r < world; ++r) {
    const float dm = gathered[(int64_t)r * rank_stride + c] / n - mean;
    Q += gathered[(int64_t)r * rank_stride + C + c] + n * dm * dm;
  }
  if (ragged) {
    S = Q = __builtin_nanf("");
    if (flag && c == mod * D) atomicOr(flag, 1);
  }
  out[c] = S;
  out[C + c] = Q;
}

// ------------------------------------------------------------------ time mean
// block = 32 float4 columns x 8 time groups; group tg sums t = tg, tg+8, ...; the 8 partials are combined in
// fixed order through LDS
__global__ __launch_bounds__(256) void mean_time_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int D) {
  __shared__ f32x4 red[8][32];
  const int b = blockIdx.x;
  const int c4 = blockIdx.y * 32 + (threadIdx.x & 31);   // float4 column
  const int tg = threadIdx.x >> 5;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 * 4 < D) {
    const float* p = x + (int64_t)b * T * D + c4 * 4;
#pragma unroll 4
    for (int t = tg; t < T; t += 8) acc += *reinterpret_cast<const f32x4*>(p + (int64_t)t * D);
  }
  red[tg][threadIdx.x & 31] = acc;
  __syncthreads();
  if (tg == 0 && c4 * 4 < D) {
    f32x4 s = red[0][threadIdx.x];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += red[i][threadIdx.x];
    *reinterpret_cast<f32x4*>(out + (int64_t)b * D + c4 * 4) = s / (float)T;
  }
}
__global__ void mean_time_fwd_scalar_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int D) {
  const int b = blockIdx.x;
  for (int d = blockIdx.y * blockDim.x + threadIdx.x; d < D; d += gridDim.y * blockDim.x) {
    float acc = 0.f;
    const float* p = x + (int64_t)b * T * D + d;
    for (int t = 0; t < T; ++t) acc += p[(int64_t)t * D];
    out[(int64_t)b * D + d] = acc / (float)T;
  }
}
__global__ void mean_time_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ dOut2, float* __restrict__ dx, int B,
                                     int T, int D, int accumulate) {
  const int64_t total = (int64_t)B * T * D;
  const float inv = 1.f / (float)T;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const int b = (int)(e / ((int64_t)T * D));
    const float g = (dOut[(int64_t)b * D + d] + (dOut2 ? dOut2[(int64_t)b * D + d] : 0.f)) * inv;
    dx[e] = accumulate ? dx[e] + g : g;
  }
}

// float4 form (D % 4 == 0, 16-byte aligned): 32-bit index arithmetic; dOut2 (optional): a second upstream gradient of the
// mean -- its two consumers' gradients are added here instead of by a separate kernel
__global__ __launch_bounds__(256) void mean_time_bwd_v4_kernel(const float* __restrict__ dOut, const float* __restrict__ dOut2,
                                                               float* __restrict__ dx, int B, int T, int D4, int accumulate) {
  const int rows = B * T;
  const float inv = 1.f / (float)T;
  const int total = rows * D4;      // < 2^31 float4s (checked by the launcher)
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int row = e / D4, d4 = e - row * D4;
    const int b = row / T;
    f32x4 g = *reinterpret_cast<const f32x4*>(dOut + ((int64_t)b * D4 + d4) * 4);
    if (dOut2) g += *reinterpret_cast<const f32x4*>(dOut2 + ((int64_t)b * D4 + d4) * 4);
    g *= inv;
    f32x4* o = reinterpret_cast<f32x4*>(dx + (int64_t)e * 4);
    *o = accumulate ? *o + g : g;
  }
}

// ------------------------------------------------------------------ narrow log-softmax / KL
__global__ void log_softmax_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int N) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = in + (int64_t)row * N;
  float m = -3.0e38f;
  for (int i = lane; i < N; i += 64) m = fmaxf(m, p[i]);
  m = wave_max(m);
  float s = 0.f;
  for (int i = lane; i < N; i += 64) s += expf(p[i] - m);
  s = wave_sum(s);
  const float lse = m + logf(s);
  for (int i = lane; i < N; i += 64) out[(int64_t)row * N + i] = p[i] - lse;
}
__global__ void log_softmax_rows_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                            float* __restrict__ dIn, int rows, int N) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < N; i += 64) s += dOut[(int64_t)row * N + i];
  s = wave_sum(s);
  for (int i = lane; i < N; i += 64) {
    const int64_t o = (int64_t)row * N + i;
    dIn[o] = dOut[o] - expf(out[o]) * s;
  }
}
__global__ void kldiv_mean_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                  float* __restrict__ loss_out, float* __restrict__ dX,
                                  const float* __restrict__ gscale, float scale, int total) {
  __shared__ float red[16];
  float acc = 0.f;
  const float inv = 1.f / (float)total;
  const float g = (gscale ? gscale[0] : 1.f) * scale;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const float tt = t[e];
    if (tt > 0.f) acc += tt * (logf(tt) - x[e]);
    if (dX) dX[e] = -tt * g * inv;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0 && loss_out) loss_out[0] = acc * inv;
}

// ------------------------------------------------------------------ criterion tail
__global__ void loss_combine_kernel(const float* __restrict__ slab, int n_terms, int stride, const float* __restrict__ coef,
                                    float* __restrict__ total, const int* __restrict__ m_dst, const int* __restrict__ m_src,
                                    const float* __restrict__ m_scale, int n_meters, float* __restrict__ meters) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;   // a dozen scalars: one lane, fixed order
  float t = 0.f;
  for (int i = 0; i < n_terms; ++i) t += coef[i] * slab[(int64_t)i * stride];
  total[0] = t;
  for (int j = 0; j < n_meters; ++j) meters[m_dst[j]] += m_scale[j] * slab[m_src[j]];
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ gtotal, const float* __restrict__ coef, int n_terms,
                                        int stride, float* __restrict__ gslab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_terms * stride) return;
  gslab[i] = (i % stride == 0) ? coef[i / stride] * gtotal[0] : 0.f;
}

// The whole criterion tail of a step in ONE launch each way (misc/crit.py:62-114,214-239 behind the fused vocabulary loss): per
// decoding pass the five criterion scalars (nacf_nll_reduce: same thread -> row map and block_sum order, the same bits), the legacy
// KLDivLoss mean of the length head (nacf_kldiv_mean: 256 threads walk the elements), then the weighted total and the running meters
// (nacf_loss_combine) -- four single-workgroup launches of 5-10 us each otherwise.
// The scalar tail (total + meters): the index / scale / coefficient / meter loads go out at the top, one per thread, under the
// reductions; the slab lives in LDS; the first thread of every meter adds that meter's entries in table order -- the same operations
// on the same values as one lane walking the table (which, measured, cost the same: the kernel's time was in the reductions' loads).
constexpr int CRIT_TAIL_SLAB = 128, CRIT_TAIL_METERS = 64;
__global__ __launch_bounds__(1024) void crit_tail_fwd_kernel(nacf_crit_tail t, float* __restrict__ slab, int n_terms, int stride,
                                                             const float* __restrict__ coef, float* __restrict__ total,
                                                             const int* __restrict__ m_dst, const int* __restrict__ m_src,
                                                             const float* __restrict__ m_scale, int n_meters, float* __restrict__ meters) {
  __shared__ float red[16], red5[16 * 5];
  __shared__ float sslab[CRIT_TAIL_SLAB], scoef[CRIT_TAIL_SLAB], mval[CRIT_TAIL_METERS], mscale[CRIT_TAIL_METERS];
  __shared__ int mdst[CRIT_TAIL_METERS];
  const int tid = threadIdx.x;
  // terms this launch does not produce keep what their producers left in the slab
  if (tid < n_terms * stride) sslab[tid] = slab[tid];
  if (tid < n_terms) scoef[tid] = coef[tid];
  int my_dst = -1, my_src = 0;
  float my_scale = 0.f, my_meter = 0.f;
  if (tid < n_meters) {
    my_dst = m_dst[tid]; my_src = m_src[tid]; my_scale = m_scale[tid];
    my_meter = meters[my_dst];
  }
  __syncthreads();
  // Every load of the reductions below is issued unconditionally, in batches of four rows / elements per thread, and all of them ahead
  // of the first block-wide sum (a load behind `if (label != PAD)` is a dependent load: three label -> log p -> argmax chains per pass and
  // ten t -> x chains of the length term were ~20 L2 latencies in a row).  The five sums of a pass share one pair of barriers.  Every sum
  // is formed from the same values in the same order as in nacf_nll_reduce / nacf_kldiv_mean.
  float part[4][5];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int k = 0; k < 5; ++k) part[pass][k] = 0.f;
    if (pass < t.n_pass) {
      const float* __restrict__ label_logp = t.label_logp[pass];
      const int64_t* __restrict__ argmax = t.argmax[pass];
      const int64_t* __restrict__ labels = t.labels[pass];
      const int exclude_mask = t.exclude[pass];
      const int rows = t.rows[pass];
      float nll = 0.f, hit = 0.f, cnt = 0.f, sl = 0.f, sc = 0.f;
      for (int r0 = tid; r0 < rows; r0 += 4 * (int)blockDim.x) {
        int64_t lab[4], am[4];
        float lp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u * (int)blockDim.x;
          const int rc = r < rows ? r : rows - 1;
          lab[u] = labels[rc]; lp[u] = label_logp[rc]; am[u] = argmax[rc];
          if (r >= rows) lab[u] = NACF_PAD;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (lab[u] != NACF_PAD) {
            nll -= lp[u]; sl += lp[u]; sc += 1.f;
            if (!(exclude_mask && lab[u] == NACF_MASK)) {
              cnt += 1.f;
              if (am[u] == lab[u]) hit += 1.f;
            }
          }
        }
      }
      part[pass][0] = nll; part[pass][1] = hit; part[pass][2] = cnt; part[pass][3] = sl; part[pass][4] = sc;
    }
  }
  float kl_acc = 0.f;
  if (t.kl_x && tid < 256)
    for (int e0 = tid; e0 < t.kl_total; e0 += 4 * 256) {
      float tt[4], xx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256;
        const int ec = e < t.kl_total ? e : t.kl_total - 1;
        tt[u] = t.kl_t[ec]; xx[u] = t.kl_x[ec];
        if (e >= t.kl_total) tt[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tt[u] > 0.f) kl_acc += tt[u] * (logf(tt[u]) - xx[u]);
    }
  // block-wide sums: wave_sum, then the wave partials in wave order (block_sum's order), five values per barrier pair
  const int w = tid >> 6, nw = ((int)blockDim.x + 63) >> 6;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (pass < t.n_pass) {
      float v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = wave_sum(part[pass][k]);
      __syncthreads();
      if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red5[w * 5 + k] = v[k];
      }
      __syncthreads();
      if (tid < 5) {
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += red5[i * 5 + tid];
        slab[(int64_t)t.slot[pass] * stride + tid] = tot;
        sslab[t.slot[pass] * stride + tid] = tot;
      }
    }
  }
  if (t.kl_x) {
    const float inv = 1.f / (float)t.kl_total;
    const float acc = block_sum(kl_acc, red);
    if (tid == 0) { slab[(int64_t)t.kl_slot * stride] = acc * inv; sslab[t.kl_slot * stride] = acc * inv; }
  }
  __syncthreads();
  if (tid < n_meters) { mdst[tid] = my_dst; mscale[tid] = my_scale; mval[tid] = sslab[my_src]; }
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int i = 0; i < n_terms; ++i) tot += scoef[i] * sslab[i * stride];
    total[0] = tot;
  }
  if (tid < n_meters) {
    bool first = true;
    for (int k = 0; k < tid; ++k) first = first && (mdst[k] != my_dst);
    if (first) {
      float m = my_meter;
      for (int k = tid; k < n_meters; ++k)
        if (mdst[k] == my_dst) m += mscale[k] * mval[k];
      meters[my_dst] = m;
    }
  }
}
__global__ __launch_bounds__(256) void crit_tail_bwd_kernel(nacf_crit_tail t, const float* __restrict__ gtotal, const float* __restrict__ coef,
                                                            int n_terms, int stride, float* __restrict__ gslab, float* __restrict__ kl_dx) {
  for (int i = threadIdx.x; i < n_terms * stride; i += blockDim.x) gslab[i] = (i % stride == 0) ? coef[i / stride] * gtotal[0] : 0.f;
  if (t.kl_t && kl_dx) {
    const float inv = 1.f / (float)t.kl_total;
    const float g = (coef[t.kl_slot] * gtotal[0]) * 1.f;      // (= gscale[0] * scale of nacf_kldiv_mean's backward form)
    for (int e = threadIdx.x; e < t.kl_total; e += blockDim.x) kl_dx[e] = -t.kl_t[e] * g * inv;
  }
}

// ------------------------------------------------------------------ epilogue backward
// (scalar form for unaligned operands; the same live-row handling as the float4 kernel below: dead rows of dY and of the
//  pre-activation may be uninitialised under the no-fill policy and must not be read -- ADVICE round 4)
__global__ void epilogue_bwd_kernel(const float* __restrict__ dY, int64_t lddy, float* __restrict__ dZ, int64_t lddz,
                                    float* __restrict__ dR, int64_t lddr, int accumulate_dR, int M, int N,
                                    nacf_epilogue ep, const int* __restrict__ rows, const int* __restrict__ count) {
  const int64_t total = (int64_t)M * N;
  const bool any_drop = ep.p_drop1 > 0.f || ep.p_drop2 > 0.f;
  const int n_live = rows ? min(M, *count) : M;
  DropRng rng;
  if (any_drop) rng.init(ep.rng_state);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(q / N), n = (int)(q % N);
    const int m = rows ? rows[pos] : pos;
    if (pos >= n_live) {                     // a dead row: its gradient is zero by the row mask
      if (dR && !accumulate_dR) dR[(int64_t)m * lddr + n] = 0.f;
      continue;
    }
    const int64_t e = (int64_t)m * N + n;    // the dropout masks are functions of the PHYSICAL element index
    float g = dY[(int64_t)m * lddy + n];
    if (ep.row_tokens && ep.row_tokens[m] == NACF_PAD) g = 0.f;
    if (ep.p_drop2 > 0.f) g *= rng.keep1((uint64_t)e, ep.salt2, ep.p_drop2);
    if (dR) {
      float* r = dR + (int64_t)m * lddr + n;
      *r = accumulate_dR ? *r + g : g;
    }
    if (ep.p_drop1 > 0.f) g *= rng.keep1((uint64_t)e, ep.salt1, ep.p_drop1);
    if (ep.act != NACF_ACT_NONE) g *= act_grad(ep.act, ep.preact[(int64_t)m * ep.ld_preact + n], n, ep.act_split);
    dZ[(int64_t)m * lddz + n] = g;
  }
}

// float4 variant (N % 4 == 0, 16-byte aligned rows): one Philox call covers the 4 elements it was drawn for.
// With a live-row list (rows / count: nacf_rowset) only the listed live rows are walked -- the GEMMs that consume dZ take the
// same list, so its dead rows are never read -- and the dead rows of dR (which flows on to consumers that read every row:
// the embedding backward) get the zeros the row mask would have produced, without reading anything.  The dropout masks are
// functions of the PHYSICAL element index, as in the forward epilogue.
__global__ void epilogue_bwd_v4_kernel(const float* __restrict__ dY, int64_t lddy, float* __restrict__ dZ, int64_t lddz,
                                       float* __restrict__ dR, int64_t lddr, int accumulate_dR, int M, int N,
                                       nacf_epilogue ep, const int* __restrict__ rows, const int* __restrict__ count) {
  const int N4 = N >> 2;
  const int64_t total = (int64_t)M * N4;
  const bool any_drop = ep.p_drop1 > 0.f || ep.p_drop2 > 0.f;
  const int n_live = rows ? min(M, *count) : M;
  DropRng rng;
  if (any_drop) rng.init(ep.rng_state);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(q / N4), n = (int)(q % N4) * 4;
    const int m = rows ? rows[pos] : pos;
    if (pos >= n_live) {                     // a dead row: its gradient is zero by the row mask
      if (dR && !accumulate_dR) *reinterpret_cast<f32x4*>(dR + (int64_t)m * lddr + n) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const int64_t e = (int64_t)m * N4 + (n >> 2);
    f32x4 g = *reinterpret_cast<const f32x4*>(dY + (int64_t)m * lddy + n);
    if (ep.row_tokens && ep.row_tokens[m] == NACF_PAD) g = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ep.p_drop2 > 0.f) g *= rng.keep4((uint64_t)e, ep.salt2, ep.p_drop2);
    if (dR) {
      f32x4* r = reinterpret_cast<f32x4*>(dR + (int64_t)m * lddr + n);
      *r = accumulate_dR ? *r + g : g;
    }
    if (ep.p_drop1 > 0.f) g *= rng.keep4((uint64_t)e, ep.salt1, ep.p_drop1);
    if (ep.act != NACF_ACT_NONE) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(ep.preact + (int64_t)m * ep.ld_preact + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] *= act_grad(ep.act, z[i], n + i, ep.act_split);
    }
    *reinterpret_cast<f32x4*>(dZ + (int64_t)m * lddz + n) = g;
  }
}

// ------------------------------------------------------------------ vocabulary log-softmax / NLL
__global__ __launch_bounds__(256) void vocab_logsoftmax_fwd_kernel(float* __restrict__ logits, int64_t ld, int V,
                                                                    const int64_t* __restrict__ labels,
                                                                    float* __restrict__ lse_out,
                                                                    int64_t* __restrict__ argmax_out,
                                                                    float* __restrict__ label_logp, int skip_pad_rows) {
  __shared__ float redv[4];
  __shared__ int redi[4];
  __shared__ float reds[16];
  const int row = blockIdx.x;
  if (skip_pad_rows && labels && labels[row] == NACF_PAD) {   // row never projected (live-row GEMM): nothing to normalise
    if (threadIdx.x == 0 && label_logp) label_logp[row] = 0.f;
    return;
  }
  float* p = logits + (int64_t)row * ld;
  float best = -3.0e38f;
  int bidx = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += 256) {
    const float v = p[i];
    if (v > best) { best = v; bidx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = best; redi[threadIdx.x >> 6] = bidx; }
  __syncthreads();
  best = redv[0]; bidx = redi[0];
  for (int w = 1; w < 4; ++w)
    if (redv[w] > best || (redv[w] == best && redi[w] < bidx)) { best = redv[w]; bidx = redi[w]; }
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) s += expf(p[i] - best);
  s = block_sum(s, reds);
  const float lse = best + logf(s);
  for (int i = threadIdx.x; i < V; i += 256) p[i] = p[i] - lse;
  if (threadIdx.x == 0) {
    if (lse_out) lse_out[row] = lse;
    if (argmax_out) argmax_out[row] = bidx;
  }
  if (labels && label_logp) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int64_t lab = labels[row];
      label_logp[row] = (lab >= 0 && lab < V) ? p[lab] : 0.f;
    }
  }
}

// Same result, ONE read of the row: the row is staged in LDS (V floats; 42 KB for V = 10547) while the maximum is
// taken, the exp-sum and the normalised write-back then come out of LDS.  HBM traffic 1 read + 1 write instead of
// 3 reads + 1 write; float4 accesses (rows are 16-byte aligned: ld % 4 == 0), scalar tail for V % 4.
__global__ __launch_bounds__(256) void vocab_logsoftmax_fwd_lds_kernel(float* __restrict__ logits, int64_t ld, int V,
                                                                        const int64_t* __restrict__ labels,
                                                                        float* __restrict__ lse_out,
                                                                        int64_t* __restrict__ argmax_out,
                                                                        float* __restrict__ label_logp, int skip_pad_rows) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  __shared__ float redv[4];
  __shared__ int redi[4];
  __shared__ float reds[16];
  const int row = blockIdx.x;
  if (skip_pad_rows && labels && labels[row] == NACF_PAD) {
    if (threadIdx.x == 0 && label_logp) label_logp[row] = 0.f;
    return;
  }
  float* p = logits + (int64_t)row * ld;
  const int V4 = V >> 2;
  float best = -3.0e38f;
  int bidx = 0x7fffffff;
  for (int i = threadIdx.x; i < V4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
    reinterpret_cast<f32x4*>(rowbuf)[i] = v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[e] > best) { best = v[e]; bidx = 4 * i + e; }
  }
  for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) {
    const float v = p[i];
    rowbuf[i] = v;
    if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = best; redi[threadIdx.x >> 6] = bidx; }
  __syncthreads();
  best = redv[0]; bidx = redi[0];
  for (int w = 1; w < 4; ++w)
    if (redv[w] > best || (redv[w] == best && redi[w] < bidx)) { best = redv[w]; bidx = redi[w]; }
  float s = 0.f;
  for (int i = threadIdx.x; i < V4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(rowbuf)[i];
    s += (expf(v[0] - best) + expf(v[1] - best)) + (expf(v[2] - best) + expf(v[3] - best));
  }
  for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) s += expf(rowbuf[i] - best);
  s = block_sum(s, reds);
  const float lse = best + logf(s);
  for (int i = threadIdx.x; i < V4; i += 256) reinterpret_cast<f32x4*>(p)[i] = reinterpret_cast<const f32x4*>(rowbuf)[i] - lse;
  for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) p[i] = rowbuf[i] - lse;
  if (threadIdx.x == 0) {
    if (lse_out) lse_out[row] = lse;
    if (argmax_out) argmax_out[row] = bidx;
    if (labels && label_logp) {
      const int64_t lab = labels[row];
      label_logp[row] = (lab >= 0 && lab < V) ? rowbuf[lab] - lse : 0.f;
    }
  }
}

__global__ void nll_reduce_kernel(const float* __restrict__ label_logp, const int64_t* __restrict__ argmax,
                                  const int64_t* __restrict__ labels, int rows, int exclude_mask,
                                  float* __restrict__ out5) {
  __shared__ float red[16];
  float nll = 0.f, hit = 0.f, cnt = 0.f, sl = 0.f, sc = 0.f;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    const int64_t lab = labels[r];
    if (lab != NACF_PAD) {
      const float lp = label_logp[r];
      nll -= lp; sl += lp; sc += 1.f;
      if (!(exclude_mask && lab == NACF_MASK)) {
        cnt += 1.f;
        if (argmax[r] == lab) hit += 1.f;
      }
    }
  }
  nll = block_sum(nll, red);
  hit = block_sum(hit, red);
  cnt = block_sum(cnt, red);
  sl = block_sum(sl, red);
  sc = block_sum(sc, red);
  if (threadIdx.x == 0) { out5[0] = nll; out5[1] = hit; out5[2] = cnt; out5[3] = sl; out5[4] = sc; }
}

// several passes that sit back to back in one [n_pass * rows, .] batch (the two NACF decoding passes): one launch, block b
// = pass b, each with its own exclusion rule and output slot
constexpr int PASS_MAX = 4;
struct NllPasses { int exclude[PASS_MAX]; float* out5[PASS_MAX]; };
__global__ void nll_reduce_multi_kernel(const float* __restrict__ label_logp, const int64_t* __restrict__ argmax,
                                        const int64_t* __restrict__ labels, int rows_per_pass, NllPasses t) {
  __shared__ float red[16];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_pass;
  const int exclude_mask = t.exclude[blockIdx.x];
  float nll = 0.f, hit = 0.f, cnt = 0.f, sl = 0.f, sc = 0.f;
  for (int r = threadIdx.x; r < rows_per_pass; r += blockDim.x) {
    const int64_t lab = labels[r0 + r];
    if (lab != NACF_PAD) {
      const float lp = label_logp[r0 + r];
      nll -= lp; sl += lp; sc += 1.f;
      if (!(exclude_mask && lab == NACF_MASK)) {
        cnt += 1.f;
        if (argmax[r0 + r] == lab) hit += 1.f;
      }
    }
  }
  nll = block_sum(nll, red);
  hit = block_sum(hit, red);
  cnt = block_sum(cnt, red);
  sl = block_sum(sl, red);
  sc = block_sum(sc, red);
  float* out5 = t.out5[blockIdx.x];
  if (threadIdx.x == 0) { out5[0] = nll; out5[1] = hit; out5[2] = cnt; out5[3] = sl; out5[4] = sc; }
}

__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logp, int64_t ld,
                                                        float* __restrict__ dlogits, int64_t ldd, int V,
                                                        const int64_t* __restrict__ labels,
                                                        const float* __restrict__ gscale, float scale,
                                                        int skip_pad_rows) {
  const int row = blockIdx.x;
  const int64_t lab = labels[row];
  if (skip_pad_rows && lab == NACF_PAD) return;   // row is outside the live-row list of the backward GEMMs
  const float g = (gscale ? gscale[0] : 1.f) * scale;
  const float* p = logp + (int64_t)row * ld;
  float* d = dlogits + (int64_t)row * ldd;
  // float4 body when both rows are 16-byte aligned (ld % 4 == 0: vocab_ld), scalar tail / fallback otherwise
  const bool vec = ((ld | ldd) & 3) == 0 && ((reinterpret_cast<uintptr_t>(logp) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
  const int V4 = vec ? (V >> 2) : 0;
  if (lab == NACF_PAD) {
    for (int i = threadIdx.x; i < V4; i += 256) reinterpret_cast<f32x4*>(d)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) d[i] = 0.f;
  } else {
    for (int i = threadIdx.x; i < V4; i += 256) {
      const f32x4 lp = reinterpret_cast<const f32x4*>(p)[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (expf(lp[e]) - ((4 * i + e) == lab ? 1.f : 0.f)) * g;
      reinterpret_cast<f32x4*>(d)[i] = o;
    }
    for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) {
      const float sm = expf(p[i]);
      d[i] = (sm - (i == lab ? 1.f : 0.f)) * g;
    }
  }
}

struct XentPasses { const float* g[4]; };
// the same from RAW logits and the row's log-sum-exp (nacf_vocab_lse_fwd): softmax = exp(z - lse)
__global__ __launch_bounds__(256) void xent_bwd_lse_kernel(const float* __restrict__ z, int64_t ld,
                                                            const float* __restrict__ lse, float* __restrict__ dlogits,
                                                            int64_t ldd, int V, const int64_t* __restrict__ labels,
                                                            const float* __restrict__ gscale, float scale,
                                                            int skip_pad_rows, int rows_per_pass, XentPasses passes) {
  const int row = blockIdx.x;
  const int64_t lab = labels[row];
  if (skip_pad_rows && lab == NACF_PAD) return;
  if (rows_per_pass > 0) gscale = passes.g[row / rows_per_pass];      // per-pass upstream gradient (passes back to back)
  const float g = (gscale ? gscale[0] : 1.f) * scale;
  const float* p = z + (int64_t)row * ld;
  float* d = dlogits + (int64_t)row * ldd;
  const bool vec = ((ld | ldd) & 3) == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
  const int V4 = vec ? (V >> 2) : 0;
  if (lab == NACF_PAD) {
    for (int i = threadIdx.x; i < V4; i += 256) reinterpret_cast<f32x4*>(d)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) d[i] = 0.f;
  } else {
    const float l = lse[row];
    for (int i = threadIdx.x; i < V4; i += 256) {
      const f32x4 zz = reinterpret_cast<const f32x4*>(p)[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (expf(zz[e] - l) - ((4 * i + e) == lab ? 1.f : 0.f)) * g;
      reinterpret_cast<f32x4*>(d)[i] = o;
    }
    for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) d[i] = (expf(p[i] - l) - (i == lab ? 1.f : 0.f)) * g;
  }
}

__global__ __launch_bounds__(256) void vocab_logsoftmax_bwd_kernel(const float* __restrict__ dlogp, int64_t ldg,
                                                                    const float* __restrict__ logp, int64_t ld,
                                                                    float* __restrict__ dlogits, int64_t ldd, int V) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const float* g = dlogp + (int64_t)row * ldg;
  const float* p = logp + (int64_t)row * ld;
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) s += g[i];
  s = block_sum(s, red);
  float* d = dlogits + (int64_t)row * ldd;
  for (int i = threadIdx.x; i < V; i += 256) d[i] = g[i] - expf(p[i]) * s;
}

// ------------------------------------------------------------------ Adam
template <bool ZERO>
__global__ void adam_step_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, const float* __restrict__ lr_p,
                                 const int64_t* __restrict__ step_p, float b1, float b2, float eps, float wd,
                                 float clip, float gscale) {
  // step_p already holds the 1-based step of THIS update (bumped by adam_bump_kernel)
  const float t = (float)step_p[0];
  const float lr = lr_p[0];
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = grad[i] * gscale;
    if (ZERO) grad[i] = 0.f;          // the next step's zero_grad, for 4 of the 32 bytes per parameter this walk moves anyway
    g = fminf(fmaxf(g, -clip), clip);
    const float p = param[i];
    g += wd * p;
    const float mm = b1 * m[i] + (1.f - b1) * g;
    const float vv = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mm;
    v[i] = vv;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    param[i] = p - step_size * (mm / denom);
  }
}
__global__ void adam_bump_kernel(int64_t* step_p) { step_p[0] += 1; }
// torch.optim.RMSprop (momentum 0, not centred: what misc/optim.py:52-60 constructs) behind the same clip / scale / zero-fill walk:
//   g += wd * p;  sq = alpha * sq + (1 - alpha) * g * g;  p -= lr * g / (sqrt(sq) + eps)
template <bool ZERO>
__global__ void rmsprop_step_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ sq, int64_t n,
                                    const float* __restrict__ lr_p, float alpha, float eps, float wd, float clip, float gscale) {
  const float lr = lr_p[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = grad[i] * gscale;
    if (ZERO) grad[i] = 0.f;
    g = fminf(fmaxf(g, -clip), clip);
    const float p = param[i];
    g += wd * p;
    const float s = alpha * sq[i] + (1.f - alpha) * g * g;
    sq[i] = s;
    param[i] = p - lr * (g / (sqrtf(s) + eps));
  }
}

inline int grid_for(int64_t total, int block = 256, int cap = 8192) {
  int64_t b = (total + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <class... P>
inline bool bn_aligned16(P... ptrs) {
  uintptr_t acc = 0;
  ((acc |= reinterpret_cast<uintptr_t>(ptrs)), ...);   // null pointers (absent operands) are fine
  return (acc & 15) == 0;
}
constexpr int BN_MAX_SLABS = 32;   // row slabs of the BatchNorm reductions (partials: [BN_MAX_SLABS][D] per statistic)
inline void bn_split(int rows, int* S, int* rows_per) {
  int s = cdiv(rows, 64);
  if (s > BN_MAX_SLABS) s = BN_MAX_SLABS;
  if (s < 1) s = 1;
  *rows_per = cdiv(rows, s);
  *S = cdiv(rows, *rows_per);
}

}  // namespace

extern "C" {

int nacf_highway_mix_fwd(const float* H, const float* TG, float* out, int rows, int D, float p_drop,
                         uint32_t salt, const uint64_t* rng_state, nacf_stream_t stream) {
  NACF_CHECK(H && TG && out && rows > 0 && D > 0, NACF_EINVAL, "nacf_highway_mix_fwd: bad argument");
  NACF_CHECK(p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL, "nacf_highway_mix_fwd: dropout needs rng_state, p<1");
  if (D % 4 == 0 && bn_aligned16(H, TG, out))
    hipLaunchKernelGGL(highway_mix_fwd_v4_kernel, dim3(grid_for((int64_t)rows * (D / 4))), dim3(256), 0, as_hip(stream), H, TG,
                       out, rows, D, p_drop, salt, rng_state);
  else
    hipLaunchKernelGGL(highway_mix_fwd_kernel, dim3(grid_for((int64_t)rows * D)), dim3(256), 0, as_hip(stream), H, TG, out,
                       rows, D, p_drop, salt, rng_state);
  NACF_LAUNCH_CHECK("nacf_highway_mix_fwd");
  return NACF_OK;
}

int nacf_highway_mix_bwd(const float* dOut, const float* H, const float* TG, float* dH, float* dP, int rows, int D,
                         float p_drop, uint32_t salt, const uint64_t* rng_state, nacf_stream_t stream) {
  NACF_CHECK(dOut && H && TG && dH && dP && rows > 0 && D > 0, NACF_EINVAL, "nacf_highway_mix_bwd: bad argument");
  NACF_CHECK(p_drop < 1.f && !(p_drop > 0.f && !rng_state), NACF_EINVAL, "nacf_highway_mix_bwd: dropout needs rng_state, p<1");
  if (D % 4 == 0 && bn_aligned16(dOut, H, TG, dH, dP))
    hipLaunchKernelGGL(highway_mix_bwd_v4_kernel, dim3(grid_for((int64_t)rows * (D / 4))), dim3(256), 0, as_hip(stream), dOut, H,
                       TG, dH, dP, rows, D, p_drop, salt, rng_state);
  else
    hipLaunchKernelGGL(highway_mix_bwd_kernel, dim3(grid_for((int64_t)rows * D)), dim3(256), 0, as_hip(stream), dOut, H, TG,
                       dH, dP, rows, D, p_drop, salt, rng_state);
  NACF_LAUNCH_CHECK("nacf_highway_mix_bwd");
  return NACF_OK;
}

size_t nacf_bn_workspace(int rows, int D) {
  (void)rows;
  return (size_t)2 * BN_MAX_SLABS * D * sizeof(float) + 256;
}

// n_mod modalities of one joint representation per launch (x[i]: [B, F[i], D] -> frames f_off[i] .. of out [B, M_total, D]).
// ws: n_mod * nacf_bn_workspace bytes.  Buffers that are not 16-byte aligned / D % 4 != 0: one modality at a time.
int nacf_bn_concat_fwd_multi(int n_mod, const float* const* x, float* out, int B, const int* F, int D, int M_total, const int* f_off,
                             const float* const* weight, const float* const* bias, float* const* running_mean,
                             float* const* running_var, int64_t* const* num_batches_tracked, float* const* save_mean,
                             float* const* save_invstd, int training, float momentum, float eps, const float* stats_global,
                             const int64_t* n_total, void* ws, size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(n_mod >= 1 && n_mod <= BN_MAX_MODS && x && out && F && f_off && B > 0 && D > 0, NACF_EINVAL, "nacf_bn_concat_fwd_multi: bad argument");
  NACF_CHECK(!stats_global || (n_total && training), NACF_EINVAL, "nacf_bn_concat_fwd_multi: global statistics need n_total and training mode");
  const size_t ws_one = nacf_bn_workspace(0, D);
  NACF_CHECK(ws && ws_bytes >= ws_one * n_mod, NACF_EWORKSPACE, "nacf_bn_concat_fwd_multi: workspace too small");
  auto at = [](auto* const* arr, int i) { return arr ? arr[i] : nullptr; };
  bool v4 = (D % 4 == 0) && bn_aligned16(out, ws);
  BnMods t = {};
  int S_max = 1;
  for (int i = 0; i < n_mod; ++i) {
    NACF_CHECK(x[i] && F[i] > 0 && f_off[i] >= 0 && f_off[i] + F[i] <= M_total, NACF_EINVAL, "nacf_bn_concat_fwd_multi: frame window outside the memory");
    NACF_CHECK(training || (at(running_mean, i) && at(running_var, i)), NACF_EINVAL, "nacf_bn_concat_fwd_multi: eval mode needs running stats");
    BnMod& m = t.m[i];
    m.x = x[i]; m.F = F[i]; m.f_off = f_off[i];
    bn_split(B * F[i], &m.S, &m.rows_per);
    m.SP = m.S; m.n_stat = (float)(B * F[i]);
    m.w = at(weight, i); m.b = at(bias, i);
    m.running_mean = at(running_mean, i); m.running_var = at(running_var, i); m.nbt = at(num_batches_tracked, i);
    m.save_mean = at(save_mean, i); m.save_invstd = at(save_invstd, i);
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_one * i);
    m.part_out0 = part; m.part_out1 = part + (size_t)BN_MAX_SLABS * D;
    m.part_in0 = m.part_out0; m.part_in1 = m.part_out1;
    if (stats_global) {           // data-parallel: (sum | squared deviations) of the GLOBAL batch, n_total[i] rows behind them
      NACF_CHECK(n_total[i] >= (int64_t)B * F[i], NACF_EINVAL, "nacf_bn_concat_fwd_multi: n_total below the local rows");
      m.SP = 1; m.n_stat = (float)n_total[i];
      m.part_in0 = stats_global + (size_t)i * D; m.part_in1 = stats_global + (size_t)(n_mod + i) * D;
    }
    v4 = v4 && bn_aligned16(m.x, m.w, m.b, m.save_mean, m.save_invstd, m.part_in0, m.part_in1);
    if (m.S > S_max) S_max = m.S;
  }
  hipStream_t s = as_hip(stream);
  if (v4) {
    dim3 grid(cdiv(D, 64), S_max, n_mod);
    if (training && !stats_global) {
      hipLaunchKernelGGL(bn_partial_sum_v4_kernel, grid, dim3(256), 0, s, t, B, D);
      hipLaunchKernelGGL(bn_partial_sqdev_v4_kernel, grid, dim3(256), 0, s, t, B, D);
    }
    hipLaunchKernelGGL(bn_apply_v4_kernel, grid, dim3(256), 0, s, t, out, B, D, M_total, training, momentum, eps);
  } else {
    for (int i = 0; i < n_mod; ++i) {
      const BnMod& m = t.m[i];
      dim3 grid(cdiv(D, 64), m.S);
      if (training && !stats_global) {
        hipLaunchKernelGGL(bn_partial_sum_kernel, grid, dim3(256), 0, s, m.x, B * m.F, D, m.rows_per, m.part_out0);
        hipLaunchKernelGGL(bn_partial_sqdev_kernel, grid, dim3(256), 0, s, m.x, B * m.F, D, m.rows_per, m.S, m.part_in0, m.part_out1, m.n_stat);
      }
      hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, s, m.x, out, B, m.F, D, M_total, m.f_off, m.w, m.b, m.running_mean,
                         m.running_var, m.nbt, m.save_mean, m.save_invstd, training, momentum, eps, m.SP, m.part_in0, m.part_in1,
                         m.rows_per, m.n_stat);
    }
  }
  NACF_LAUNCH_CHECK("nacf_bn_concat_fwd_multi");
  return NACF_OK;
}

int nacf_bn_concat_fwd(const float* x, float* out, int B, int F, int D, int M_total, int f_off, const float* weight,
                       const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                       float* save_mean, float* save_invstd, int training, float momentum, float eps, void* ws,
                       size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(x && out && B > 0 && F > 0 && D > 0, NACF_EINVAL, "nacf_bn_concat_fwd: bad argument");
  return nacf_bn_concat_fwd_multi(1, &x, out, B, &F, D, M_total, &f_off, &weight, &bias, &running_mean, &running_var,
                                  &num_batches_tracked, &save_mean, &save_invstd, training, momentum, eps, nullptr, nullptr, ws,
                                  ws_bytes, stream);
}

int nacf_bn_concat_bwd_multi(int n_mod, const float* dOut, const float* const* x, float* const* dx, int B, const int* F, int D,
                             int M_total, const int* f_off, const float* const* weight, const float* const* save_mean,
                             const float* const* save_invstd, float* const* dweight, float* const* dbias, float beta,
                             const float* sums_global, const int64_t* n_total, void* ws, size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(n_mod >= 1 && n_mod <= BN_MAX_MODS && dOut && x && dx && F && f_off && save_mean && save_invstd && B > 0 && D > 0,
             NACF_EINVAL, "nacf_bn_concat_bwd_multi: bad argument");
  NACF_CHECK(!sums_global || n_total, NACF_EINVAL, "nacf_bn_concat_bwd_multi: global sums need n_total");
  const size_t ws_one = nacf_bn_workspace(0, D);
  NACF_CHECK(ws && ws_bytes >= ws_one * n_mod, NACF_EWORKSPACE, "nacf_bn_concat_bwd_multi: workspace too small");
  auto at = [](auto* const* arr, int i) { return arr ? arr[i] : nullptr; };
  bool v4 = (D % 4 == 0) && bn_aligned16(dOut, ws);
  BnMods t = {};
  int S_max = 1;
  for (int i = 0; i < n_mod; ++i) {
    NACF_CHECK(x[i] && dx[i] && save_mean[i] && save_invstd[i], NACF_EINVAL, "nacf_bn_concat_bwd_multi: null pointer");
    NACF_CHECK(F[i] > 0 && f_off[i] >= 0 && f_off[i] + F[i] <= M_total, NACF_EINVAL, "nacf_bn_concat_bwd_multi: bad shape");
    BnMod& m = t.m[i];
    m.x = x[i]; m.dx = dx[i]; m.F = F[i]; m.f_off = f_off[i];
    bn_split(B * F[i], &m.S, &m.rows_per);
    m.SP = m.S; m.n_stat = (float)(B * F[i]);
    m.w = at(weight, i);
    m.save_mean = const_cast<float*>(save_mean[i]); m.save_invstd = const_cast<float*>(save_invstd[i]);
    m.dweight = at(dweight, i); m.dbias = at(dbias, i);
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_one * i);
    m.part_out0 = part; m.part_out1 = part + (size_t)BN_MAX_SLABS * D;
    m.part_in0 = m.part_out0; m.part_in1 = m.part_out1;
    if (sums_global) {            // data-parallel: [n_mod][2][D] = (sum dy | sum dy * xhat) of the GLOBAL batch; the LOCAL
                                  // parameter gradients were written by nacf_bn_sync_bwd_local_multi
      NACF_CHECK(n_total[i] >= (int64_t)B * F[i], NACF_EINVAL, "nacf_bn_concat_bwd_multi: n_total below the local rows");
      m.SP = 1; m.n_stat = (float)n_total[i];
      m.part_in0 = sums_global + (size_t)(2 * i) * D; m.part_in1 = sums_global + (size_t)(2 * i + 1) * D;
      m.dweight = nullptr; m.dbias = nullptr;
    }
    v4 = v4 && bn_aligned16(m.x, m.dx, m.w, m.save_mean, m.save_invstd, m.part_in0, m.part_in1);
    if (m.S > S_max) S_max = m.S;
  }
  hipStream_t s = as_hip(stream);
  if (v4) {
    dim3 grid(cdiv(D, 64), S_max, n_mod);
    if (!sums_global) hipLaunchKernelGGL(bn_bwd_partial_v4_kernel, grid, dim3(256), 0, s, t, dOut, B, D, M_total);
    hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, grid, dim3(256), 0, s, t, dOut, B, D, M_total, beta);
  } else {
    for (int i = 0; i < n_mod; ++i) {
      const BnMod& m = t.m[i];
      dim3 grid(cdiv(D, 64), m.S);
      if (!sums_global)
        hipLaunchKernelGGL(bn_bwd_partial_kernel, grid, dim3(256), 0, s, dOut, m.x, B, m.F, D, M_total, m.f_off, m.save_mean,
                           m.save_invstd, m.rows_per, m.part_out0, m.part_out1);
      hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, dOut, m.x, m.dx, B, m.F, D, M_total, m.f_off, m.w, m.save_mean,
                         m.save_invstd, m.dweight, m.dbias, beta, m.SP, m.rows_per, m.part_in0, m.part_in1, m.n_stat);
    }
  }
  NACF_LAUNCH_CHECK("nacf_bn_concat_bwd_multi");
  return NACF_OK;
}

int nacf_bn_concat_bwd(const float* dOut, const float* x, float* dx, int B, int F, int D, int M_total, int f_off,
                       const float* weight, const float* save_mean, const float* save_invstd, float* dweight,
                       float* dbias, float beta, void* ws, size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(dOut && x && dx && save_mean && save_invstd, NACF_EINVAL, "nacf_bn_concat_bwd: null pointer");
  return nacf_bn_concat_bwd_multi(1, dOut, &x, &dx, B, &F, D, M_total, &f_off, &weight, &save_mean, &save_invstd, &dweight, &dbias,
                                  beta, nullptr, nullptr, ws, ws_bytes, stream);
}

// ---- data-parallel (synchronised) BatchNorm: the same two-pass statistics over the GLOBAL batch, cut where the ranks
// exchange a [D] vector (the caller all-reduces between the calls; nothing here knows about processes)
int nacf_bn_sync_stat(const float* x, int rows, int D, const float* sum_global, int64_t n_total, float* out, void* ws,
                      size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(x && out && rows > 0 && D > 0 && n_total >= rows, NACF_EINVAL, "nacf_bn_sync_stat: bad argument");
  NACF_CHECK(ws && ws_bytes >= nacf_bn_workspace(rows, D), NACF_EWORKSPACE, "nacf_bn_sync_stat: workspace too small");
  int S, rows_per;
  bn_split(rows, &S, &rows_per);
  float* part = reinterpret_cast<float*>(ws);
  hipStream_t s = as_hip(stream);
  dim3 grid(cdiv(D, 64), S);
  const bool v4 = (D % 4 == 0) && bn_aligned16(x, sum_global, ws);
  BnMods t = {};
  BnMod& m = t.m[0];
  m.x = x; m.F = rows; m.rows_per = rows_per; m.S = S; m.SP = 1; m.n_stat = (float)n_total;     // B = 1, F = rows
  m.part_in0 = sum_global; m.part_out0 = part; m.part_out1 = part;
  if (!sum_global) {
    if (v4) hipLaunchKernelGGL(bn_partial_sum_v4_kernel, grid, dim3(256), 0, s, t, 1, D);
    else hipLaunchKernelGGL(bn_partial_sum_kernel, grid, dim3(256), 0, s, x, rows, D, rows_per, part);
  } else {            // squared deviations from the GLOBAL mean = sum_global / n_total (one "slab" of partial sums)
    if (v4) hipLaunchKernelGGL(bn_partial_sqdev_v4_kernel, grid, dim3(256), 0, s, t, 1, D);
    else hipLaunchKernelGGL(bn_partial_sqdev_kernel, grid, dim3(256), 0, s, x, rows, D, rows_per, 1, sum_global, part, (float)n_total);
  }
  hipLaunchKernelGGL(bn_fold_parts_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, part, S, D, out, (float*)nullptr, 0.f);
  NACF_LAUNCH_CHECK("nacf_bn_sync_stat");
  return NACF_OK;
}

// this rank's share of the forward statistics for every modality: loc [2][n_mod][D] = (sum | squared deviations about the
// rank's OWN mean) over its B * F[i] rows -- three launches; what the ranks all-gather for nacf_bn_sync_merge
int nacf_bn_sync_local_multi(int n_mod, const float* const* x, int B, const int* F, int D, float* loc, void* ws, size_t ws_bytes,
                             nacf_stream_t stream) {
  NACF_CHECK(n_mod >= 1 && n_mod <= BN_MAX_MODS && x && F && loc && B > 0 && D > 0, NACF_EINVAL, "nacf_bn_sync_local_multi: bad argument");
  const size_t ws_one = nacf_bn_workspace(0, D);
  NACF_CHECK(ws && ws_bytes >= ws_one * n_mod, NACF_EWORKSPACE, "nacf_bn_sync_local_multi: workspace too small");
  bool v4 = (D % 4 == 0) && bn_aligned16(ws);
  BnMods t = {};
  BnFold f = {};
  int S_max = 1;
  for (int i = 0; i < n_mod; ++i) {
    NACF_CHECK(x[i] && F[i] > 0, NACF_EINVAL, "nacf_bn_sync_local_multi: bad modality");
    BnMod& m = t.m[i];
    m.x = x[i]; m.F = F[i];
    bn_split(B * F[i], &m.S, &m.rows_per);
    m.SP = m.S; m.n_stat = (float)(B * F[i]);
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_one * i);
    m.part_out0 = part; m.part_out1 = part + (size_t)BN_MAX_SLABS * D;
    m.part_in0 = m.part_out0; m.part_in1 = m.part_out1;
    f.part[i][0] = m.part_out0; f.part[i][1] = m.part_out1; f.S[i] = m.S;
    f.out[i][0] = loc + (size_t)i * D; f.out[i][1] = loc + (size_t)(n_mod + i) * D;
    v4 = v4 && bn_aligned16(m.x);
    if (m.S > S_max) S_max = m.S;
  }
  hipStream_t s = as_hip(stream);
  if (v4) {
    dim3 grid(cdiv(D, 64), S_max, n_mod);
    hipLaunchKernelGGL(bn_partial_sum_v4_kernel, grid, dim3(256), 0, s, t, B, D);
    hipLaunchKernelGGL(bn_partial_sqdev_v4_kernel, grid, dim3(256), 0, s, t, B, D);
  } else {
    for (int i = 0; i < n_mod; ++i) {
      const BnMod& m = t.m[i];
      dim3 grid(cdiv(D, 64), m.S);
      hipLaunchKernelGGL(bn_partial_sum_kernel, grid, dim3(256), 0, s, m.x, B * m.F, D, m.rows_per, m.part_out0);
      hipLaunchKernelGGL(bn_partial_sqdev_kernel, grid, dim3(256), 0, s, m.x, B * m.F, D, m.rows_per, m.S, m.part_in0, m.part_out1, m.n_stat);
    }
  }
  hipLaunchKernelGGL(bn_fold_multi_kernel, dim3(cdiv(D, 256), n_mod, 2), dim3(256), 0, s, f, D, 0.f);
  NACF_LAUNCH_CHECK("nacf_bn_sync_local_multi");
  return NACF_OK;
}

// this rank's share of the backward sums for every modality: sums [n_mod][2][D] = (sum dy | sum dy * xhat) over its rows,
// also accumulated (beta) into the LOCAL dbias | dweight -- two launches; the ranks all-reduce `sums`
int nacf_bn_sync_bwd_local_multi(int n_mod, const float* dOut, const float* const* x, int B, const int* F, int D, int M_total,
                                 const int* f_off, const float* const* save_mean, const float* const* save_invstd, float* sums,
                                 float* const* dweight, float* const* dbias, float beta, void* ws, size_t ws_bytes,
                                 nacf_stream_t stream) {
  NACF_CHECK(n_mod >= 1 && n_mod <= BN_MAX_MODS && dOut && x && F && f_off && save_mean && save_invstd && sums && B > 0 && D > 0,
             NACF_EINVAL, "nacf_bn_sync_bwd_local_multi: bad argument");
  const size_t ws_one = nacf_bn_workspace(0, D);
  NACF_CHECK(ws && ws_bytes >= ws_one * n_mod, NACF_EWORKSPACE, "nacf_bn_sync_bwd_local_multi: workspace too small");
  auto at = [](auto* const* arr, int i) { return arr ? arr[i] : nullptr; };
  bool v4 = (D % 4 == 0) && bn_aligned16(dOut, ws);
  BnMods t = {};
  BnFold f = {};
  int S_max = 1;
  for (int i = 0; i < n_mod; ++i) {
    NACF_CHECK(x[i] && save_mean[i] && save_invstd[i] && F[i] > 0 && f_off[i] >= 0 && f_off[i] + F[i] <= M_total, NACF_EINVAL,
               "nacf_bn_sync_bwd_local_multi: bad modality");
    BnMod& m = t.m[i];
    m.x = x[i]; m.F = F[i]; m.f_off = f_off[i];
    bn_split(B * F[i], &m.S, &m.rows_per);
    m.save_mean = const_cast<float*>(save_mean[i]); m.save_invstd = const_cast<float*>(save_invstd[i]);
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_one * i);
    m.part_out0 = part; m.part_out1 = part + (size_t)BN_MAX_SLABS * D;
    f.part[i][0] = m.part_out0; f.part[i][1] = m.part_out1; f.S[i] = m.S;
    f.out[i][0] = sums + (size_t)(2 * i) * D; f.out[i][1] = sums + (size_t)(2 * i + 1) * D;
    f.acc[i][0] = at(dbias, i); f.acc[i][1] = at(dweight, i);
    v4 = v4 && bn_aligned16(m.x, m.save_mean, m.save_invstd);
    if (m.S > S_max) S_max = m.S;
  }
  hipStream_t s = as_hip(stream);
  if (v4) {
    hipLaunchKernelGGL(bn_bwd_partial_v4_kernel, dim3(cdiv(D, 64), S_max, n_mod), dim3(256), 0, s, t, dOut, B, D, M_total);
  } else {
    for (int i = 0; i < n_mod; ++i) {
      const BnMod& m = t.m[i];
      hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(D, 64), m.S), dim3(256), 0, s, dOut, m.x, B, m.F, D, M_total, m.f_off,
                         m.save_mean, m.save_invstd, m.rows_per, m.part_out0, m.part_out1);
    }
  }
  hipLaunchKernelGGL(bn_fold_multi_kernel, dim3(cdiv(D, 256), n_mod, 2), dim3(256), 0, s, f, D, beta);
  NACF_LAUNCH_CHECK("nacf_bn_sync_bwd_local_multi");
  return NACF_OK;
}

int nacf_bn_sync_merge(const float* gathered, int world, int n_mod, int D, const float* rows_per_rank, int64_t rank_stride, float* out,
                       int32_t* ragged_flag, nacf_stream_t stream) {
  NACF_CHECK(gathered && out && rows_per_rank && world >= 1 && n_mod >= 1 && n_mod <= BN_MAX_MODS && D > 0, NACF_EINVAL,
             "nacf_bn_sync_merge: bad argument");
  BnMergeN rows = {};
  for (int i = 0; i < n_mod; ++i) {
    NACF_CHECK(rows_per_rank[i] >= 1.f, NACF_EINVAL, "nacf_bn_sync_merge: a modality without rows");
    rows.n[i] = rows_per_rank[i];
  }
  NACF_CHECK(rank_stride >= 2 * (int64_t)n_mod * D, NACF_EINVAL, "nacf_bn_sync_merge: rank stride shorter than the statistics");
  hipLaunchKernelGGL(bn_sync_merge_kernel, dim3(cdiv(n_mod * D, 256)), dim3(256), 0, as_hip(stream), gathered, world, n_mod, D, rows,
                     rank_stride, out, ragged_flag);
  NACF_LAUNCH_CHECK("nacf_bn_sync_merge");
  return NACF_OK;
}

int nacf_bn_concat_fwd_sync(const float* x, float* out, int B, int F, int D, int M_total, int f_off, const float* weight,
                            const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                            float* save_mean, float* save_invstd, float momentum, float eps, const float* sum_global,
                            const float* sqdev_global, int64_t n_total, nacf_stream_t stream) {
  NACF_CHECK(x && out && sum_global && sqdev_global && B > 0 && F > 0 && D > 0 && n_total >= (int64_t)B * F, NACF_EINVAL,
             "nacf_bn_concat_fwd_sync: bad argument");
  NACF_CHECK(f_off >= 0 && f_off + F <= M_total, NACF_EINVAL, "nacf_bn_concat_fwd_sync: frame window outside the memory");
  const int rows = B * F;
  int S, rows_per;
  bn_split(rows, &S, &rows_per);
  hipStream_t s = as_hip(stream);
  dim3 grid(cdiv(D, 64), S);
  if ((D % 4 == 0) && bn_aligned16(x, out, weight, bias, save_mean, save_invstd, sum_global, sqdev_global)) {
    BnMods t = {};
    BnMod& m = t.m[0];
    m.x = x; m.F = F; m.f_off = f_off; m.rows_per = rows_per; m.S = S; m.SP = 1; m.n_stat = (float)n_total;
    m.w = weight; m.b = bias; m.running_mean = running_mean; m.running_var = running_var; m.nbt = num_batches_tracked;
    m.save_mean = save_mean; m.save_invstd = save_invstd; m.part_in0 = sum_global; m.part_in1 = sqdev_global;
    hipLaunchKernelGGL(bn_apply_v4_kernel, grid, dim3(256), 0, s, t, out, B, D, M_total, 1, momentum, eps);
  } else
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, s, x, out, B, F, D, M_total, f_off, weight, bias, running_mean,
                       running_var, num_batches_tracked, save_mean, save_invstd, 1, momentum, eps, 1, sum_global, sqdev_global,
                       rows_per, (float)n_total);
  NACF_LAUNCH_CHECK("nacf_bn_concat_fwd_sync");
  return NACF_OK;
}

int nacf_bn_sync_bwd_stat(const float* dOut, const float* x, int B, int F, int D, int M_total, int f_off,
                          const float* save_mean, const float* save_invstd, float* sums2, float* dweight, float* dbias,
                          float beta, void* ws, size_t ws_bytes, nacf_stream_t stream) {
  NACF_CHECK(dOut && x && save_mean && save_invstd && sums2, NACF_EINVAL, "nacf_bn_sync_bwd_stat: null pointer");
  NACF_CHECK(B > 0 && F > 0 && D > 0 && f_off >= 0 && f_off + F <= M_total, NACF_EINVAL, "nacf_bn_sync_bwd_stat: bad shape");
  NACF_CHECK(ws && ws_bytes >= nacf_bn_workspace(B * F, D), NACF_EWORKSPACE, "nacf_bn_sync_bwd_stat: workspace too small");
  const int rows = B * F;
  int S, rows_per;
  bn_split(rows, &S, &rows_per);
  float* part_dy = reinterpret_cast<float*>(ws);
  float* part_dyx = part_dy + (size_t)BN_MAX_SLABS * D;
  hipStream_t s = as_hip(stream);
  dim3 grid(cdiv(D, 64), S);
  if ((D % 4 == 0) && bn_aligned16(dOut, x, save_mean, save_invstd, ws)) {
    BnMods t = {};
    BnMod& m = t.m[0];
    m.x = x; m.F = F; m.f_off = f_off; m.rows_per = rows_per; m.S = S;
    m.save_mean = const_cast<float*>(save_mean); m.save_invstd = const_cast<float*>(save_invstd);
    m.part_out0 = part_dy; m.part_out1 = part_dyx;
    hipLaunchKernelGGL(bn_bwd_partial_v4_kernel, grid, dim3(256), 0, s, t, dOut, B, D, M_total);
  } else
    hipLaunchKernelGGL(bn_bwd_partial_kernel, grid, dim3(256), 0, s, dOut, x, B, F, D, M_total, f_off, save_mean,
                       save_invstd, rows_per, part_dy, part_dyx);
  // the LOCAL sums are this rank's bias / weight gradient (the gradient all-reduce adds the ranks up); the vectors
  // in sums2 = [sum dy | sum dy*xhat] are what the ranks exchange for dx
  hipLaunchKernelGGL(bn_fold_parts_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, part_dy, S, D, sums2, dbias, beta);
  hipLaunchKernelGGL(bn_fold_parts_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, part_dyx, S, D, sums2 + D, dweight, beta);
  NACF_LAUNCH_CHECK("nacf_bn_sync_bwd_stat");
  return NACF_OK;
}

int nacf_bn_concat_bwd_sync(const float* dOut, const float* x, float* dx, int B, int F, int D, int M_total, int f_off,
                            const float* weight, const float* save_mean, const float* save_invstd, const float* sums2_global,
                            int64_t n_total, nacf_stream_t stream) {
  NACF_CHECK(dOut && x && dx && save_mean && save_invstd && sums2_global, NACF_EINVAL, "nacf_bn_concat_bwd_sync: null pointer");
  NACF_CHECK(B > 0 && F > 0 && D > 0 && f_off >= 0 && f_off + F <= M_total && n_total >= (int64_t)B * F, NACF_EINVAL,
             "nacf_bn_concat_bwd_sync: bad shape");
  const int rows = B * F;
  int S, rows_per;
  bn_split(rows, &S, &rows_per);
  hipStream_t s = as_hip(stream);
  dim3 grid(cdiv(D, 64), S);
  if ((D % 4 == 0) && bn_aligned16(dOut, x, dx, weight, save_mean, save_invstd, sums2_global)) {
    BnMods t = {};
    BnMod& m = t.m[0];
    m.x = x; m.dx = dx; m.F = F; m.f_off = f_off; m.rows_per = rows_per; m.S = S; m.SP = 1; m.n_stat = (float)n_total;
    m.w = weight; m.save_mean = const_cast<float*>(save_mean); m.save_invstd = const_cast<float*>(save_invstd);
    m.part_in0 = sums2_global; m.part_in1 = sums2_global + D;
    hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, grid, dim3(256), 0, s, t, dOut, B, D, M_total, 0.f);
  } else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, dOut, x, dx, B, F, D, M_total, f_off, weight, save_mean,
                       save_invstd, (float*)nullptr, (float*)nullptr, 0.f, 1, rows_per, sums2_global, sums2_global + D,
                       (float)n_total);
  NACF_LAUNCH_CHECK("nacf_bn_concat_bwd_sync");
  return NACF_OK;
}

int nacf_mean_time_fwd(const float* x, float* out, int B, int T, int D, nacf_stream_t stream) {
  NACF_CHECK(x && out && B > 0 && T > 0 && D > 0, NACF_EINVAL, "nacf_mean_time_fwd: bad argument");
  if (D % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
    hipLaunchKernelGGL(mean_time_fwd_kernel, dim3(B, cdiv(D, 128)), dim3(256), 0, as_hip(stream), x, out, T, D);
  else
    hipLaunchKernelGGL(mean_time_fwd_scalar_kernel, dim3(B, cdiv(D, 256)), dim3(256), 0, as_hip(stream), x, out, T, D);
  NACF_LAUNCH_CHECK("nacf_mean_time_fwd");
  return NACF_OK;
}

int nacf_mean_time_bwd(const float* dOut, const float* dOut2, float* dx, int B, int T, int D, int accumulate, nacf_stream_t stream) {
  NACF_CHECK(dOut && dx && B > 0 && T > 0 && D > 0, NACF_EINVAL, "nacf_mean_time_bwd: bad argument");
  if (D % 4 == 0 && bn_aligned16(dOut, dOut2, dx) && (int64_t)B * T * (D / 4) < (int64_t)1 << 31)
    hipLaunchKernelGGL(mean_time_bwd_v4_kernel, dim3(grid_for((int64_t)B * T * (D / 4))), dim3(256), 0, as_hip(stream), dOut, dOut2,
                       dx, B, T, D / 4, accumulate);
  else
    hipLaunchKernelGGL(mean_time_bwd_kernel, dim3(grid_for((int64_t)B * T * D)), dim3(256), 0, as_hip(stream), dOut, dOut2, dx, B,
                       T, D, accumulate);
  NACF_LAUNCH_CHECK("nacf_mean_time_bwd");
  return NACF_OK;
}

int nacf_log_softmax_rows(const float* in, float* out, int rows, int N, nacf_stream_t stream) {
  NACF_CHECK(in && out && rows > 0 && N > 0, NACF_EINVAL, "nacf_log_softmax_rows: bad argument");
  hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_hip(stream), in, out, rows, N);
  NACF_LAUNCH_CHECK("nacf_log_softmax_rows");
  return NACF_OK;
}

int nacf_log_softmax_rows_bwd(const float* dOut, const float* out, float* dIn, int rows, int N, nacf_stream_t stream) {
  NACF_CHECK(dOut && out && dIn && rows > 0 && N > 0, NACF_EINVAL, "nacf_log_softmax_rows_bwd: bad argument");
  hipLaunchKernelGGL(log_softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_hip(stream), dOut, out, dIn, rows, N);
  NACF_LAUNCH_CHECK("nacf_log_softmax_rows_bwd");
  return NACF_OK;
}

int nacf_kldiv_mean(const float* x, const float* t, float* loss_out, float* dX, const float* gscale, float scale,
                    int rows, int N, nacf_stream_t stream) {
  NACF_CHECK(x && t && (loss_out || dX) && rows > 0 && N > 0, NACF_EINVAL, "nacf_kldiv_mean: bad argument");
  hipLaunchKernelGGL(kldiv_mean_kernel, dim3(1), dim3(256), 0, as_hip(stream), x, t, loss_out, dX, gscale, scale,
                     rows * N);
  NACF_LAUNCH_CHECK("nacf_kldiv_mean");
  return NACF_OK;
}

int nacf_loss_combine(const float* slab, int n_terms, int stride, const float* coef, float* total,
                      const int32_t* m_dst, const int32_t* m_src, const float* m_scale, int n_meters,
                      float* meters, nacf_stream_t stream) {
  NACF_CHECK(slab && coef && total && n_terms > 0 && stride > 0, NACF_EINVAL, "nacf_loss_combine: bad argument");
  NACF_CHECK(n_meters == 0 || (m_dst && m_src && m_scale && meters), NACF_EINVAL, "nacf_loss_combine: incomplete meter table");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, as_hip(stream), slab, n_terms, stride, coef, total, m_dst,
                     m_src, m_scale, n_meters, meters);
  NACF_LAUNCH_CHECK("nacf_loss_combine");
  return NACF_OK;
}

int nacf_loss_combine_bwd(const float* gtotal, const float* coef, int n_terms, int stride, float* gslab,
                          nacf_stream_t stream) {
  NACF_CHECK(gtotal && coef && gslab && n_terms > 0 && stride > 0, NACF_EINVAL, "nacf_loss_combine_bwd: bad argument");
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(cdiv(n_terms * stride, 64)), dim3(64), 0, as_hip(stream), gtotal, coef,
                     n_terms, stride, gslab);
  NACF_LAUNCH_CHECK("nacf_loss_combine_bwd");
  return NACF_OK;
}

int nacf_crit_tail_fwd(const nacf_crit_tail* tail, float* slab, int n_terms, int stride, const float* coef, float* total,
                       const int32_t* m_dst, const int32_t* m_src, const float* m_scale, int n_meters, float* meters,
                       nacf_stream_t stream) {
  NACF_CHECK(tail && slab && coef && total && n_terms > 0 && stride >= 5, NACF_EINVAL, "nacf_crit_tail_fwd: bad argument");
  NACF_CHECK(n_meters == 0 || (m_dst && m_src && m_scale && meters), NACF_EINVAL, "nacf_crit_tail_fwd: incomplete meter table");
  NACF_CHECK(tail->n_pass >= 0 && tail->n_pass <= 4, NACF_EINVAL, "nacf_crit_tail_fwd: at most 4 passes");
  NACF_CHECK(n_terms * stride <= CRIT_TAIL_SLAB && n_meters <= CRIT_TAIL_METERS, NACF_EINVAL,
             "nacf_crit_tail_fwd: at most %d slab floats and %d meter entries", CRIT_TAIL_SLAB, CRIT_TAIL_METERS);
  for (int i = 0; i < tail->n_pass; ++i)
    NACF_CHECK(tail->label_logp[i] && tail->argmax[i] && tail->labels[i] && tail->rows[i] > 0 && tail->slot[i] >= 0 && tail->slot[i] < n_terms,
               NACF_EINVAL, "nacf_crit_tail_fwd: bad pass %d", i);
  NACF_CHECK(!tail->kl_x || (tail->kl_t && tail->kl_total > 0 && tail->kl_slot >= 0 && tail->kl_slot < n_terms), NACF_EINVAL,
             "nacf_crit_tail_fwd: bad length term");
  hipLaunchKernelGGL(crit_tail_fwd_kernel, dim3(1), dim3(1024), 0, as_hip(stream), *tail, slab, n_terms, stride, coef, total, m_dst,
                     m_src, m_scale, n_meters, meters);
  NACF_LAUNCH_CHECK("nacf_crit_tail_fwd");
  return NACF_OK;
}

int nacf_crit_tail_bwd(const nacf_crit_tail* tail, const float* gtotal, const float* coef, int n_terms, int stride, float* gslab,
                       float* kl_dx, nacf_stream_t stream) {
  NACF_CHECK(tail && gtotal && coef && gslab && n_terms > 0 && stride > 0, NACF_EINVAL, "nacf_crit_tail_bwd: bad argument");
  NACF_CHECK(!kl_dx || (tail->kl_t && tail->kl_total > 0 && tail->kl_slot >= 0 && tail->kl_slot < n_terms), NACF_EINVAL,
             "nacf_crit_tail_bwd: bad length term");
  hipLaunchKernelGGL(crit_tail_bwd_kernel, dim3(1), dim3(256), 0, as_hip(stream), *tail, gtotal, coef, n_terms, stride, gslab, kl_dx);
  NACF_LAUNCH_CHECK("nacf_crit_tail_bwd");
  return NACF_OK;
}

int nacf_epilogue_bwd(const float* dY, int64_t lddy, float* dZ, int64_t lddz, float* dR, int64_t lddr,
                      int accumulate_dR, int M, int N, const nacf_epilogue* ep, const nacf_rowset* rs, nacf_stream_t stream) {
  NACF_CHECK(dY && dZ && ep && M > 0 && N > 0, NACF_EINVAL, "nacf_epilogue_bwd: bad argument");
  NACF_CHECK(!rs || (rs->rows && rs->count), NACF_EINVAL, "nacf_epilogue_bwd: incomplete row set");
  NACF_CHECK(!(ep->act != NACF_ACT_NONE && !ep->preact), NACF_EINVAL, "nacf_epilogue_bwd: activation backward needs preact");
  NACF_CHECK(!((ep->p_drop1 > 0.f || ep->p_drop2 > 0.f) && !ep->rng_state), NACF_EINVAL,
             "nacf_epilogue_bwd: dropout needs rng_state");
  const bool v4 = (N % 4 == 0) && (lddy % 4 == 0) && (lddz % 4 == 0) && (!dR || lddr % 4 == 0) &&
                  (!ep->preact || ep->ld_preact % 4 == 0) && bn_aligned16(dY, dZ, dR, ep->preact);
  if (v4)
    hipLaunchKernelGGL(epilogue_bwd_v4_kernel, dim3(grid_for((int64_t)M * (N / 4))), dim3(256), 0, as_hip(stream), dY, lddy,
                       dZ, lddz, dR, lddr, accumulate_dR, M, N, *ep, rs ? rs->rows : nullptr, rs ? rs->count : nullptr);
  else
    hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(grid_for((int64_t)M * N)), dim3(256), 0, as_hip(stream), dY, lddy, dZ, lddz,
                       dR, lddr, accumulate_dR, M, N, *ep, rs ? rs->rows : nullptr, rs ? rs->count : nullptr);
  NACF_LAUNCH_CHECK("nacf_epilogue_bwd");
  return NACF_OK;
}

int nacf_vocab_logsoftmax_fwd(float* logits, int64_t ld, int rows, int V, const int64_t* labels, float* lse,
                              int64_t* argmax, float* label_logp, int skip_pad_rows, nacf_stream_t stream) {
  NACF_CHECK(logits && rows > 0 && V > 0 && ld >= V, NACF_EINVAL, "nacf_vocab_logsoftmax_fwd: bad argument");
  const size_t row_bytes = (size_t)V * sizeof(float);
  if (row_bytes <= 60 * 1024 && ld % 4 == 0 && aligned16(logits))
    hipLaunchKernelGGL(vocab_logsoftmax_fwd_lds_kernel, dim3(rows), dim3(256), (row_bytes + 15) & ~(size_t)15, as_hip(stream),
                       logits, ld, V, labels, lse, argmax, label_logp, skip_pad_rows);
  else
    hipLaunchKernelGGL(vocab_logsoftmax_fwd_kernel, dim3(rows), dim3(256), 0, as_hip(stream), logits, ld, V, labels, lse,
                       argmax, label_logp, skip_pad_rows);
  NACF_LAUNCH_CHECK("nacf_vocab_logsoftmax_fwd");
  return NACF_OK;
}

int nacf_nll_reduce(const float* label_logp, const int64_t* argmax, const int64_t* labels, int rows, int exclude_mask,
                    float* out5, nacf_stream_t stream) {
  NACF_CHECK(label_logp && argmax && labels && out5 && rows > 0, NACF_EINVAL, "nacf_nll_reduce: bad argument");
  hipLaunchKernelGGL(nll_reduce_kernel, dim3(1), dim3(1024), 0, as_hip(stream), label_logp, argmax, labels, rows,
                     exclude_mask, out5);
  NACF_LAUNCH_CHECK("nacf_nll_reduce");
  return NACF_OK;
}

int nacf_xent_bwd(const float* logp, int64_t ld, float* dlogits, int64_t ldd, int rows, int V, const int64_t* labels,
                  const float* gscale, float scale, int skip_pad_rows, nacf_stream_t stream) {
  NACF_CHECK(logp && dlogits && labels && rows > 0 && V > 0, NACF_EINVAL, "nacf_xent_bwd: bad argument");
  hipLaunchKernelGGL(xent_bwd_kernel, dim3(rows), dim3(256), 0, as_hip(stream), logp, ld, dlogits, ldd, V, labels, gscale,
                     scale, skip_pad_rows);
  NACF_LAUNCH_CHECK("nacf_xent_bwd");
  return NACF_OK;
}

int nacf_xent_bwd_lse(const float* logits, int64_t ld, const float* lse, float* dlogits, int64_t ldd, int rows, int V,
                      const int64_t* labels, const float* gscale, float scale, int skip_pad_rows, nacf_stream_t stream) {
  NACF_CHECK(logits && lse && dlogits && labels && rows > 0 && V > 0, NACF_EINVAL, "nacf_xent_bwd_lse: bad argument");
  hipLaunchKernelGGL(xent_bwd_lse_kernel, dim3(rows), dim3(256), 0, as_hip(stream), logits, ld, lse, dlogits, ldd, V, labels,
                     gscale, scale, skip_pad_rows, 0, XentPasses{});
  NACF_LAUNCH_CHECK("nacf_xent_bwd_lse");
  return NACF_OK;
}

int nacf_xent_bwd_lse_multi(const float* logits, int64_t ld, const float* lse, float* dlogits, int64_t ldd, int rows_per_pass,
                            int n_pass, int V, const int64_t* labels, const float* const* gscales, float scale, int skip_pad_rows,
                            nacf_stream_t stream) {
  NACF_CHECK(logits && lse && dlogits && labels && gscales && rows_per_pass > 0 && n_pass >= 1 && n_pass <= 4 && V > 0, NACF_EINVAL,
             "nacf_xent_bwd_lse_multi: bad argument");
  XentPasses p = {};
  for (int i = 0; i < n_pass; ++i) {
    NACF_CHECK(gscales[i], NACF_EINVAL, "nacf_xent_bwd_lse_multi: every pass needs its upstream gradient");
    p.g[i] = gscales[i];
  }
  hipLaunchKernelGGL(xent_bwd_lse_kernel, dim3(rows_per_pass * n_pass), dim3(256), 0, as_hip(stream), logits, ld, lse, dlogits, ldd, V,
                     labels, (const float*)nullptr, scale, skip_pad_rows, rows_per_pass, p);
  NACF_LAUNCH_CHECK("nacf_xent_bwd_lse_multi");
  return NACF_OK;
}

int nacf_nll_reduce_multi(const float* label_logp, const int64_t* argmax, const int64_t* labels, int rows_per_pass, int n_pass,
                          const int* exclude_mask, float* const* out5, nacf_stream_t stream) {
  NACF_CHECK(label_logp && argmax && labels && exclude_mask && out5 && rows_per_pass > 0 && n_pass >= 1 && n_pass <= PASS_MAX, NACF_EINVAL,
             "nacf_nll_reduce_multi: bad argument");
  NllPasses t = {};
  for (int i = 0; i < n_pass; ++i) {
    NACF_CHECK(out5[i], NACF_EINVAL, "nacf_nll_reduce_multi: null output");
    t.exclude[i] = exclude_mask[i];
    t.out5[i] = out5[i];
  }
  hipLaunchKernelGGL(nll_reduce_multi_kernel, dim3(n_pass), dim3(1024), 0, as_hip(stream), label_logp, argmax, labels, rows_per_pass, t);
  NACF_LAUNCH_CHECK("nacf_nll_reduce_multi");
  return NACF_OK;
}

int nacf_vocab_logsoftmax_bwd(const float* dlogp, int64_t ldg, const float* logp, int64_t ld, float* dlogits,
                              int64_t ldd, int rows, int V, nacf_stream_t stream) {
  NACF_CHECK(dlogp && logp && dlogits && rows > 0 && V > 0, NACF_EINVAL, "nacf_vocab_logsoftmax_bwd: bad argument");
  hipLaunchKernelGGL(vocab_logsoftmax_bwd_kernel, dim3(rows), dim3(256), 0, as_hip(stream), dlogp, ldg, logp, ld, dlogits,
                     ldd, V);
  NACF_LAUNCH_CHECK("nacf_vocab_logsoftmax_bwd");
  return NACF_OK;
}

int nacf_adam_step_part(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr,
                        int64_t* step_count, float beta1, float beta2, float eps, float weight_decay, float grad_clip,
                        float grad_scale, int bump, nacf_stream_t stream) {
  NACF_CHECK(param && grad && exp_avg && exp_avg_sq && lr && step_count && n > 0, NACF_EINVAL,
             "nacf_adam_step: bad argument");
  hipStream_t s = as_hip(stream);
  if (bump & 1) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, s, step_count);
  if (bump & 2)
    hipLaunchKernelGGL(adam_step_kernel<true>, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n,
                       lr, step_count, beta1, beta2, eps, weight_decay, grad_clip, grad_scale);
  else
    hipLaunchKernelGGL(adam_step_kernel<false>, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n,
                       lr, step_count, beta1, beta2, eps, weight_decay, grad_clip, grad_scale);
  NACF_LAUNCH_CHECK("nacf_adam_step");
  return NACF_OK;
}

int nacf_rmsprop_step(float* param, float* grad, float* square_avg, int64_t n, const float* lr, float alpha, float eps,
                      float weight_decay, float grad_clip, float grad_scale, int zero_grad, nacf_stream_t stream) {
  NACF_CHECK(param && grad && square_avg && lr && n > 0, NACF_EINVAL, "nacf_rmsprop_step: bad argument");
  hipStream_t s = as_hip(stream);
  if (zero_grad)
    hipLaunchKernelGGL(rmsprop_step_kernel<true>, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, param, grad, square_avg, n, lr, alpha, eps,
                       weight_decay, grad_clip, grad_scale);
  else
    hipLaunchKernelGGL(rmsprop_step_kernel<false>, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, param, grad, square_avg, n, lr, alpha, eps,
                       weight_decay, grad_clip, grad_scale);
  NACF_LAUNCH_CHECK("nacf_rmsprop_step");
  return NACF_OK;
}

int nacf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr,
                   int64_t* step_count, float beta1, float beta2, float eps, float weight_decay, float grad_clip,
                   float grad_scale, nacf_stream_t stream) {
  return nacf_adam_step_part(param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n, lr, step_count, beta1, beta2, eps, weight_decay,
                             grad_clip, grad_scale, 1, stream);
}

}  // extern "C"

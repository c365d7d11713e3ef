// Shared device/host helpers for libnacf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/nacf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- errors
void nacf_set_error(const char* fmt, ...);

#define NACF_CHECK(cond, code, ...)                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      nacf_set_error(__VA_ARGS__);                                             \
      return (code);                                                           \
    }                                                                          \
  } while (0)

#define NACF_LAUNCH_CHECK(name)                                                \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      nacf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
      return NACF_ELAUNCH;                                                     \
    }                                                                          \
  } while (0)

static inline hipStream_t as_hip(nacf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- RNG
// Philox4x32-10 counter-based generator.  counter = (element_index/4, salt,
// step_lo, step_hi), key = seed.  One call yields the 4 draws of an aligned
// group of 4 consecutive elements, so forward (vector epilogue) and backward
// (elementwise kernels) regenerate identical masks from the element index.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#ifdef NACF_PHILOX_MULHI
    // (the wide GEMM unit: 64-bit products raise the epilogue's register pressure past what hipcc fits beside the hand-allocated
    //  accumulators -- it parked spills in a0..a3, tools/check_wide_hazards.py -- so that unit keeps the two-instruction form)
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x, hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
#else
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: the quarter-rate
    // multiplies are most of a call, and a GEMM epilogue with dropout makes 16 calls per lane and 128 x 128 tile.  Same bits.
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#endif
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

struct DropRng {
  uint2 key;
  uint32_t step_lo, step_hi;
  __device__ __forceinline__ void init(const uint64_t* state) {
    uint64_t seed = state[0], step = state[1];
    key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    step_lo = (uint32_t)step;
    step_hi = (uint32_t)(step >> 32);
  }
  // keep-scale factors (0 or 1/(1-p)) for elements 4*grp .. 4*grp+3
  __device__ __forceinline__ f32x4 keep4(uint64_t grp, uint32_t salt, float p) const {
    uint4 r = philox4x32_10(make_uint4((uint32_t)grp, salt ^ ((uint32_t)(grp >> 32) * 0x9E3779B9u), step_lo, step_hi), key);
    const float inv = 1.0f / (1.0f - p);
    const float s = 1.0f / 16777216.0f;
    f32x4 o;
    o[0] = ((r.x >> 8) * s >= p) ? inv : 0.f;
    o[1] = ((r.y >> 8) * s >= p) ? inv : 0.f;
    o[2] = ((r.z >> 8) * s >= p) ? inv : 0.f;
    o[3] = ((r.w >> 8) * s >= p) ? inv : 0.f;
    return o;
  }
  // keep-scale factor of a single element index e
  __device__ __forceinline__ float keep1(uint64_t e, uint32_t salt, float p) const {
    f32x4 k = keep4(e >> 2, salt, p);
    return k[(int)(e & 3)];
  }
};

// ---------------------------------------------------------------- math
__device__ __forceinline__ float gelu_new_f(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), models/bert.py:12-13
  const float c = 0.7978845608028654f;
  float u = c * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_new_grad(float x) {
  const float c = 0.7978845608028654f;
  float x2 = x * x;
  float u = c * (x + 0.044715f * x * x2);
  float t = tanhf(u);
  float du = c * (1.0f + 3.0f * 0.044715f * x2);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  return x * 0.5f * (1.0f + erff(x * 0.7071067811865475f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float inv_sqrt2pi = 0.3989422804014327f;
  return 0.5f * (1.0f + erff(x * 0.7071067811865475f)) + x * inv_sqrt2pi * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float apply_act(int act, float z, int n, int split) {
  switch (act) {
    case NACF_ACT_RELU: return z > 0.f ? z : 0.f;
    case NACF_ACT_GELU_NEW: return gelu_new_f(z);
    case NACF_ACT_TANH: return tanhf(z);
    case NACF_ACT_SIGMOID: return sigmoid_f(z);
    case NACF_ACT_TANH_SIGMOID: return n < split ? tanhf(z) : sigmoid_f(z);
    case NACF_ACT_GELU_ERF: return gelu_erf_f(z);
    default: return z;
  }
}
__device__ __forceinline__ float act_grad(int act, float z, int n, int split) {
  switch (act) {
    case NACF_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case NACF_ACT_GELU_NEW: return gelu_new_grad(z);
    case NACF_ACT_TANH: { float t = tanhf(z); return 1.f - t * t; }
    case NACF_ACT_SIGMOID: { float s = sigmoid_f(z); return s * (1.f - s); }
    case NACF_ACT_TANH_SIGMOID: {
      if (n < split) { float t = tanhf(z); return 1.f - t * t; }
      float s = sigmoid_f(z); return s * (1.f - s);
    }
    case NACF_ACT_GELU_ERF: return gelu_erf_grad(z);
    default: return 1.f;
  }
}

// ---------------------------------------------------------------- reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Register-resident multi-head attention on the CDNA4 matrix cores (fp32 operands on v_mfma_f32_16x16x4_f32, or -- throughput
// mode -- bf16 operands on v_mfma_f32_16x16x16_bf16: "PR" below), forward and backward.  One WAVE owns one
// (sequence, head) pair: Lq <= 32 queries (2 MFMA row tiles), Lk <= 16*NKT keys,
// dk = 16*DK16.  Nothing is staged through LDS in the forward pass: the MFMA
// fragment layouts are chosen so that every operand is either a coalesced
// global load or already sitting in the accumulator registers:
//
//  S^T = K Q^T   : a = K[key=i][d], b = Q[q=i][d] with the reduce index d
//                  k-permuted (lane group g walks d = g*dk/4 + step), so a lane
//                  fetches dk/4 CONTIGUOUS floats of its own K / Q row.
//                  Result: lane (i, g) holds S[q = 16*tm + i][key = 16*tn + 4g + rr].
//  softmax       : per q row = 32 values in-lane + 2 xor-shuffles across g.
//  O = P V       : reduce index = key, k-permuted so that lane group g consumes
//                  exactly the keys whose P it already holds (b operand = the
//                  accumulator register itself); a = V[key][DK16*i + td], i.e.
//                  each lane loads DK16 contiguous floats of a V row and the
//                  output-column permutation d = DK16*n + td makes the lane's
//                  results DK16 contiguous floats of O again.
//
// Backward re-computes P the same way, gets dP = dO V^T and dQ = dS K with the
// two forward contractions, and only the reductions over q (dK = dS^T Q,
// dV = P^T dO) need a transpose, done through a wave-private LDS tile.
#pragma once
#include "common.hpp"

namespace attn {

template <int N> struct Ld;
template <> struct Ld<1> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = p[0]; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[1]) { p[0] = v[0]; }
};
template <> struct Ld<2> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) {
    const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[2]) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  }
};
template <> struct Ld<4> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};

// Arithmetic of the contractions, template parameter PR of everything below:
//   PR = 0: fp32 operands, v_mfma_f32_16x16x4_f32 (the exact and the fp32 GEMM modes);
//   PR = 1: the throughput mode (NACF_GEMM_BF16): operands rounded to bf16 (nearest even) in registers right before the matrix
//           instruction, v_mfma_f32_16x16x16_bf16 -- ONE instruction where the fp32 form issues four (a lane's four k-steps are the four
//           bf16 values of its operand: the fragment layouts above are unchanged), 16 against 128 cycles of the matrix pipe.
//           Soft-max, its statistics, dS and every accumulator stay fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 bf16x4_rne(float a, float b, float c, float d) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 w = {__builtin_bit_cast(unsigned int, __builtin_convertvector(f2{a, b}, b2)),
                __builtin_bit_cast(unsigned int, __builtin_convertvector(f2{c, d}, b2))};
  return __builtin_bit_cast(s16x4, w);
}
__device__ __forceinline__ s16x4 bf16x4_rne(const f32x4& v) { return bf16x4_rne(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ f32x4 mfma_bf16(s16x4 a, s16x4 b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}

// Guarded loads (`row < n ? load : 0`) are basic blocks of their own: the compiler never batches loads across blocks, and each then costs
// a full L2 latency.  The branch-free forms read the last live row / key instead and replace the value.  Measured per kernel inside the
// captured step and the decode loop (profiles/r06_attn_branchfree_ab.txt): it pays in the forward kernel's key-token loads (decode:
// 37.8 -> 35.4 us) and, marginally, in the dK / dV operand loads of the backward; in the row fragments it costs the self-attention
// backward 2.8 us for 1 us of the forward, and branch-free key-token loads make that backward 8 us slower.  Build switches for the A/B:
#ifndef NACF_ATTN_BF_ROWFRAG
#define NACF_ATTN_BF_ROWFRAG 0       // load_row_frag
#endif
#ifndef NACF_ATTN_BF_KEYTOK
#define NACF_ATTN_BF_KEYTOK 1        // softmax_rows<.., FWD = true> (the forward kernels); 2 = the backward too
#endif
#ifndef NACF_ATTN_BF_CQ
#define NACF_ATTN_BF_CQ 1            // contract_q
#endif
#ifndef NACF_ATTN_KEYTOK_BALLOT
#define NACF_ATTN_KEYTOK_BALLOT 1    // softmax_rows: the key tokens as one load per wave + ballot (supersedes NACF_ATTN_BF_KEYTOK)
#endif
// row fragments of a [rows, dk] operand for the "reduce over d" contractions:
// lane (i, g) loads floats [g*dk/4, (g+1)*dk/4) of row (16*t + i); rows >= n_rows read as zero
template <int DK16>
__device__ __forceinline__ void load_row_frag(f32x4 (&f)[DK16], const float* __restrict__ base, int64_t ld, int row,
                                              int n_rows, int g) {
  constexpr int DK = 16 * DK16;
#if NACF_ATTN_BF_ROWFRAG
  // a guarded load is a basic block of its own, and loads in different blocks are never batched: every fragment then costs one full
  // L2 latency.  Read the last live row instead (n_rows >= 1) and replace the value by zero.
  const bool live = row < n_rows;
  const float* src = base + (int64_t)(live ? row : n_rows - 1) * ld + g * (DK / 4);
#pragma unroll
  for (int j = 0; j < DK16; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * j);
    f[j] = live ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#else
#pragma unroll
  for (int j = 0; j < DK16; ++j) {
    f[j] = (row < n_rows) ? *reinterpret_cast<const f32x4*>(base + (int64_t)row * ld + g * (DK / 4) + 4 * j)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#endif
}

// acc[tm][tn] (+)= sum_d  A[16*tn + i][d] * B[16*tm + i][d]   (A: keys side, B: queries side)
template <int NKT, int DK16, int PR = 0>
__device__ __forceinline__ void contract_d(f32x4 (&acc)[2][NKT], const float* __restrict__ A, int64_t lda, int n_a,
                                           const f32x4 (&bf)[2][DK16], int i, int g) {
  s16x4 bp[2][DK16];
  if constexpr (PR == 1) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int j = 0; j < DK16; ++j) bp[tm][j] = bf16x4_rne(bf[tm][j]);
  }
#pragma unroll
  for (int tn = 0; tn < NKT; ++tn) {
    f32x4 af[DK16];
    load_row_frag<DK16>(af, A, lda, tn * 16 + i, n_a, g);
#pragma unroll
    for (int j = 0; j < DK16; ++j) {
      if constexpr (PR == 1) {
        const s16x4 a = bf16x4_rne(af[j]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm][tn] = mfma_bf16(a, bp[tm][j], acc[tm][tn]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][e], bf[tm][j][e], acc[tm][tn], 0, 0, 0);
      }
    }
  }
}

// o[tm][td] = sum_key  P[16*tm + i][key] * A[key][DK16*n + td]   with P in the accumulator layout of contract_d
template <int NKT, int DK16, int PR = 0>
__device__ __forceinline__ void contract_key(f32x4 (&o)[2][DK16], const f32x4 (&p)[2][NKT],
                                             const float* __restrict__ A, int64_t lda, int n_a, int i, int g) {
  if constexpr (PR == 1) {
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn) {
      float af[4][DK16];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        int key = tn * 16 + g * 4 + rr;
        key = key < n_a ? key : n_a - 1;          // P is exactly 0 there; keep the address in range
        Ld<DK16>::ld(A + (int64_t)key * lda + DK16 * i, af[rr]);
      }
      const s16x4 pp[2] = {bf16x4_rne(p[0][tn]), bf16x4_rne(p[1][tn])};
#pragma unroll
      for (int td = 0; td < DK16; ++td) {
        const s16x4 a = bf16x4_rne(af[0][td], af[1][td], af[2][td], af[3][td]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) o[tm][td] = mfma_bf16(a, pp[tm], o[tm][td]);
      }
    }
    return;
  }
#pragma unroll
  for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      int key = tn * 16 + g * 4 + rr;
      key = key < n_a ? key : n_a - 1;          // P is exactly 0 there; keep the address in range
      float af[DK16];
      Ld<DK16>::ld(A + (int64_t)key * lda + DK16 * i, af);
#pragma unroll
      for (int td = 0; td < DK16; ++td)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          o[tm][td] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[td], p[tm][tn][rr], o[tm][td], 0, 0, 0);
    }
}

// store o[tm][td] (layout of contract_key) as rows of a [rows, dk] matrix
template <int DK16>
__device__ __forceinline__ void store_rows(const f32x4 (&o)[2][DK16], float* __restrict__ out, int64_t ld, int n_rows,
                                           int i, int g) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int q = tm * 16 + i;
    if (q < n_rows) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float v[DK16];
#pragma unroll
        for (int td = 0; td < DK16; ++td) v[td] = o[tm][td][rr];
        Ld<DK16>::st(out + (int64_t)q * ld + DK16 * (g * 4 + rr), v);
      }
    }
  }
}

// scores (accumulator layout) -> probabilities, in place.  Mirrors models/bert.py:157-167:
// divide by sqrt(dk) AFTER the product, masked_fill(-10e6), softmax over keys.
template <int NKT, bool FWD = false>
__device__ __forceinline__ void softmax_rows(f32x4 (&s)[2][NKT], float sq, const int64_t* __restrict__ key_tok,
                                             int causal, int Lk, int i, int g) {
  unsigned long long padbits = 0ull;  // bit (tn*4+rr): key is PAD or beyond Lk handled separately
  if (key_tok) {
#if NACF_ATTN_KEYTOK_BALLOT
    // one token per lane, one load for the whole wave; the <pad> flags travel as a ballot (the 8 or 32 keys of a lane used to be
    // 8 or 32 guarded loads, each behind the one before it)
    const int lane = threadIdx.x & 63;
    const unsigned long long m0 = __ballot(lane < Lk && key_tok[lane < Lk ? lane : 0] == NACF_PAD);
    unsigned long long m1 = 0ull;
    if constexpr (NKT > 4) m1 = __ballot(lane + 64 < Lk && key_tok[lane + 64 < Lk ? lane + 64 : 0] == NACF_PAD);
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int key = tn * 16 + g * 4 + rr;
        const unsigned long long bit = tn < 4 ? (m0 >> key) & 1ull : (m1 >> (key - 64)) & 1ull;
        padbits |= bit << (tn * 4 + rr);
      }
#else
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int key = tn * 16 + g * 4 + rr;
        if constexpr (NACF_ATTN_BF_KEYTOK > (FWD ? 0 : 1)) {
          const int64_t tok = key_tok[key < Lk ? key : Lk - 1];     // one batch of loads, not a guarded load (= a latency) per key
          if (key < Lk && tok == NACF_PAD) padbits |= 1ull << (tn * 4 + rr);
        } else {
          if (key < Lk && key_tok[key] == NACF_PAD) padbits |= 1ull << (tn * 4 + rr);
        }
      }
#endif
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int q = tm * 16 + i;
    float mx = -3.0e38f;
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int key = tn * 16 + g * 4 + rr;
        float v = s[tm][tn][rr] / sq;
        if ((padbits >> (tn * 4 + rr)) & 1ull) v = -10e6f;
        if (causal && (key > q || (causal > 1 && key <= q - (causal - 1)))) v = -10e6f;     // causal - 1 = --watch window
        if (key >= Lk) v = -3.0e38f;             // tile padding: contributes exactly 0
        s[tm][tn][rr] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float e = expf(s[tm][tn][rr] - mx);
        s[tm][tn][rr] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) s[tm][tn][rr] = s[tm][tn][rr] / sum;
  }
}

// One item = (sequence, head, block of 32 queries), the work of one wave; nqb = ceil(Lq / 32) > 1 only without a causal mask.
// Shared by fwd_kernel (one item per wave of the grid) and by persistent callers (a workgroup's
// waves walk the items of an attention stage).
template <int NKT, int DK16, int PR = 0>
__device__ __forceinline__ void fwd_item(const int item, const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                         int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                         float* __restrict__ O, int64_t ldo,
                                         const int64_t* __restrict__ key_tokens, int causal,
                                         float* __restrict__ probs, int R, int H, int Lq, int Lk, int kv_div,
                                         int kv_mod, int nqb) {
  constexpr int DK = 16 * DK16;
  const int lane = threadIdx.x & 63;
  const int qb = item % nqb, rh = item / nqb;
  const int r = rh / H, h = rh % H;
  const int q0 = qb * 32;
  const int nq = min(32, Lq - q0);
  const int kvr = (r / kv_div) % kv_mod;
  const int i = lane & 15, g = lane >> 4;
  const float* Qb = Q + ((int64_t)r * Lq + q0) * ldq + h * DK;
  const float* Kb = K + (int64_t)kvr * Lk * ldk + h * DK;
  const float* Vb = V + (int64_t)kvr * Lk * ldv + h * DK;
  f32x4 qf[2][DK16];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) load_row_frag<DK16>(qf[tm], Qb, ldq, tm * 16 + i, nq, g);
  f32x4 s[2][NKT];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn) s[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  contract_d<NKT, DK16, PR>(s, Kb, ldk, Lk, qf, i, g);
  softmax_rows<NKT, true>(s, sqrtf((float)DK), key_tokens ? key_tokens + (int64_t)r * Lk : nullptr, causal, Lk, i, g);
  if (probs) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int q = tm * 16 + i;
      if (q < nq) {
#pragma unroll
        for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int key = tn * 16 + g * 4 + rr;
            if (key < Lk) probs[(((int64_t)h * R + r) * Lq + q0 + q) * Lk + key] = s[tm][tn][rr];
          }
      }
    }
  }
  f32x4 o[2][DK16];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int td = 0; td < DK16; ++td) o[tm][td] = f32x4{0.f, 0.f, 0.f, 0.f};
  contract_key<NKT, DK16, PR>(o, s, Vb, ldv, Lk, i, g);
  store_rows<DK16>(o, O + ((int64_t)r * Lq + q0) * ldo + h * DK, ldo, nq, i, g);
}

template <int NKT, int DK16, int PR = 0>
__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                   int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                                   float* __restrict__ O, int64_t ldo,
                                                   const int64_t* __restrict__ key_tokens, int causal,
                                                   float* __restrict__ probs, int R, int H, int Lq, int Lk, int kv_div,
                                                   int kv_mod, int nqb) {
  // The blocks of one (sequence, head) are consecutive items, i.e. waves of ONE workgroup: they walk the same K / V rows together.
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= R * H * nqb) return;
  fwd_item<NKT, DK16, PR>(item, Q, ldq, K, ldk, V, ldv, O, ldo, key_tokens, causal, probs, R, H, Lq, Lk, kv_div, kv_mod, nqb);
}

// dst[key][d] (+)= sum_q T[q][key] * A[q][d]  for key tiles [tk0, tk0+TKC), T = wave-private LDS tile [32][PITCH]
template <int NKT, int DK16, int TKC, int PR = 0>
__device__ __forceinline__ void contract_q(const float* T, int pitch, const float* __restrict__ A,
                                           int64_t lda, int n_q, float* __restrict__ dst, int64_t ldd, int n_keys,
                                           bool accumulate, int i, int g) {
#pragma unroll
  for (int tk0 = 0; tk0 < NKT; tk0 += TKC) {
    f32x4 acc[DK16][TKC];
#pragma unroll
    for (int td = 0; td < DK16; ++td)
#pragma unroll
      for (int t = 0; t < TKC; ++t) acc[td][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PR == 1) {
#pragma unroll
      for (int st0 = 0; st0 < 8; st0 += 4) {       // four k-steps of the fp32 form = one bf16 instruction
        float af[4][DK16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = (st0 + e) * 4 + g;
          const bool live = q < n_q;
          Ld<DK16>::ld(A + (int64_t)(live ? q : n_q - 1) * lda + DK16 * i, af[e]);
#pragma unroll
          for (int td = 0; td < DK16; ++td) af[e][td] = live ? af[e][td] : 0.f;
        }
        s16x4 a[DK16];
#pragma unroll
        for (int td = 0; td < DK16; ++td) a[td] = bf16x4_rne(af[0][td], af[1][td], af[2][td], af[3][td]);
#pragma unroll
        for (int t = 0; t < TKC; ++t) {
          const int c = (tk0 + t) * 16 + i;
          const s16x4 b = bf16x4_rne(T[(st0 * 4 + g) * pitch + c], T[((st0 + 1) * 4 + g) * pitch + c],
                                     T[((st0 + 2) * 4 + g) * pitch + c], T[((st0 + 3) * 4 + g) * pitch + c]);
#pragma unroll
          for (int td = 0; td < DK16; ++td) acc[td][t] = mfma_bf16(a[td], b, acc[td][t]);
        }
      }
    } else {
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int q = st * 4 + g;                 // k-permutation of the reduce index q (Lq padded to 32)
      float af[DK16];
#if NACF_ATTN_BF_CQ
      const bool live = q < n_q;
      Ld<DK16>::ld(A + (int64_t)(live ? q : n_q - 1) * lda + DK16 * i, af);
#pragma unroll
      for (int td = 0; td < DK16; ++td) af[td] = live ? af[td] : 0.f;
#else
#pragma unroll
      for (int td = 0; td < DK16; ++td) af[td] = 0.f;
      if (q < n_q) Ld<DK16>::ld(A + (int64_t)q * lda + DK16 * i, af);
#endif
#pragma unroll
      for (int t = 0; t < TKC; ++t) {
        const float b = T[q * pitch + (tk0 + t) * 16 + i];
#pragma unroll
        for (int td = 0; td < DK16; ++td)
          acc[td][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[td], b, acc[td][t], 0, 0, 0);
      }
    }
    }
#pragma unroll
    for (int t = 0; t < TKC; ++t) {
      const int key = (tk0 + t) * 16 + i;
      if (key < n_keys) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float* p = dst + (int64_t)key * ldd + DK16 * (g * 4 + rr);
          float v[DK16];
          if (accumulate) Ld<DK16>::ld(p, v);
#pragma unroll
          for (int td = 0; td < DK16; ++td) v[td] = accumulate ? v[td] + acc[td][t][rr] : acc[td][t][rr];
          Ld<DK16>::st(p, v);
        }
      }
    }
  }
}

template <int NKT>
__device__ __forceinline__ void tile_to_lds(float* T, int pitch, const f32x4 (&p)[2][NKT], int i, int g) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
      *reinterpret_cast<f32x4*>(&T[(tm * 16 + i) * pitch + tn * 16 + g * 4]) = p[tm][tn];
}

// Backward.  An "item" is one (key/value row set, head): all sequences that attend to the same memory (both NACF passes
// of a video; one sequence for self-attention) contribute to the same dK / dV.  `wpi` waves share an item: wave `sub`
// takes the sequences k = sub, sub + wpi, ... of the item.  Everything up to dQ runs in parallel; the dK / dV updates
// (read-modify-write of the same global rows) are then applied one wave at a time in sequence order, separated by
// workgroup barriers -- deterministic, and with two sequences per video twice the waves are in flight.
template <int NKT, int DK16, int PR = 0>
__global__ __launch_bounds__(256) void bwd_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                   int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                                   const float* __restrict__ dO, int64_t lddo, float* __restrict__ dQ,
                                                   int64_t lddq, float* __restrict__ dK, int64_t lddk,
                                                   float* __restrict__ dV, int64_t lddv,
                                                   const int64_t* __restrict__ key_tokens, int causal, int R, int n_kv,
                                                   int H, int Lq, int Lk, int kv_div, int kv_mod, int wpi, int rounds) {
  constexpr int DK = 16 * DK16;
  constexpr int PITCH = NKT * 16 + 16;          // = 16 mod 32 dwords: the q = 4*st + g rows of a read land 16 banks apart
  constexpr int TKC = NKT < 4 ? NKT : 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* T = smem + wave * 32 * PITCH;
  const int item = (blockIdx.x * 4 + wave) / wpi, sub = wave % wpi;
  const bool active = item < n_kv * H;          // inactive waves still take part in the barriers
  const int kvr = active ? item / H : 0, h = active ? item % H : 0;
  const int i = lane & 15, g = lane >> 4;
  const float sq = sqrtf((float)DK);
  const float* Kb = K + (int64_t)kvr * Lk * ldk + h * DK;
  const float* Vb = V + (int64_t)kvr * Lk * ldv + h * DK;
  float* dKb = dK + (int64_t)kvr * Lk * lddk + h * DK;
  float* dVb = dV + (int64_t)kvr * Lk * lddv + h * DK;
  for (int round = 0; round < rounds; ++round) {
    // the k-th sequence of this memory row set: r = (q * kv_mod + kvr) * kv_div + j with k = q * kv_div + j
    const int k = round * wpi + sub;
    const int r = ((k / kv_div) * kv_mod + kvr) * kv_div + k % kv_div;
    const bool has = active && r < R && (k / kv_div) * kv_mod + kvr < (R + kv_div - 1) / kv_div;
    const float* Qb = Q + (int64_t)(has ? r : 0) * Lq * ldq + h * DK;
    const float* dOb = dO + (int64_t)(has ? r : 0) * Lq * lddo + h * DK;
    f32x4 p[2][NKT], dp[2][NKT];
    if (has) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < NKT; ++tn) { p[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      {
        f32x4 qf[2][DK16];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) load_row_frag<DK16>(qf[tm], Qb, ldq, tm * 16 + i, Lq, g);
        contract_d<NKT, DK16, PR>(p, Kb, ldk, Lk, qf, i, g);
      }
      softmax_rows<NKT>(p, sq, key_tokens ? key_tokens + (int64_t)r * Lk : nullptr, causal, Lk, i, g);
      {
        f32x4 gf[2][DK16];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) load_row_frag<DK16>(gf[tm], dOb, lddo, tm * 16 + i, Lq, g);
        contract_d<NKT, DK16, PR>(dp, Vb, ldv, Lk, gf, i, g);   // dP = dO V^T
      }
      // dS = P * (dP - rowsum(P * dP)) / sqrt(dk)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        float dot = 0.f;
#pragma unroll
        for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) dot += p[tm][tn][rr] * dp[tm][tn][rr];
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) dp[tm][tn][rr] = p[tm][tn][rr] * (dp[tm][tn][rr] - dot) / sq;
      }
      {
        f32x4 o[2][DK16];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int td = 0; td < DK16; ++td) o[tm][td] = f32x4{0.f, 0.f, 0.f, 0.f};
        contract_key<NKT, DK16, PR>(o, dp, Kb, ldk, Lk, i, g);   // dQ = dS K
        store_rows<DK16>(o, dQ + (int64_t)r * Lq * lddq + h * DK, lddq, Lq, i, g);
      }
    }
    // reductions over q: transpose dS / P through the wave-private LDS tile; one wave of the item at a time
    for (int turn = 0; turn < wpi; ++turn) {
      if (has && turn == sub) {
        const bool seen = k > 0;
        tile_to_lds<NKT>(T, PITCH, dp, i, g);
        contract_q<NKT, DK16, TKC, PR>(T, PITCH, Qb, ldq, Lq, dKb, lddk, Lk, seen, i, g);    // dK (+)= dS^T Q
        tile_to_lds<NKT>(T, PITCH, p, i, g);
        contract_q<NKT, DK16, TKC, PR>(T, PITCH, dOb, lddo, Lq, dVb, lddv, Lk, seen, i, g);  // dV (+)= P^T dO
      }
      if (wpi > 1) __syncthreads();
    }
  }
}

// ---- key-block backward (cross-attention: no key mask, not causal, up to 128 keys) ------------------------------------
// The kernel above gives one wave a whole (memory row set, head): 1024 waves for B=128 x 8 heads -- one per SIMD, each
// walking 1280 MFMAs per sequence behind uncovered load latency.  Here the FOUR waves of a workgroup share the item and
// split its keys into blocks of 32: a wave computes S, P, dP, dS only for its block, owns the dK / dV rows of that block
// (accumulated in registers over the item's sequences: no read-modify-write of global rows, no ordering between
// sequences), and the three quantities that span all keys are combined through LDS in a fixed wave order:
//   row max / row sum of exp  (softmax statistics, flash-style rescaling),
//   delta = rowsum(P * dP),
//   dQ = sum over key blocks of dS_blk K_blk.
// Branch-free forms of the fragment loads (rows past the end: the last live row is read and the value replaced by zero).  A guarded
// load is a basic block of its own: the compiler cannot batch such loads, and a contraction that takes its operand from global memory
// step by step then waits out one full L2 latency per step (contract_q_acc: 16 of them per sequence).  n_rows >= 1.
template <int DK16>
__device__ __forceinline__ void raw_row_frag(f32x4 (&f)[DK16], const float* __restrict__ base, int64_t ld, int row, int n_rows, int g) {
  constexpr int DK = 16 * DK16;
  const float* src = base + (int64_t)(row < n_rows ? row : n_rows - 1) * ld + g * (DK / 4);
#pragma unroll
  for (int j = 0; j < DK16; ++j) f[j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
}
template <int DK16>
__device__ __forceinline__ void mask_row_frag(f32x4 (&f)[DK16], bool live) {
#pragma unroll
  for (int j = 0; j < DK16; ++j) f[j] = live ? f[j] : f32x4{0.f, 0.f, 0.f, 0.f};
}
// acc[tm][tn] += sum_d A[16 tn + i][d] B[16 tm + i][d] with both operands in registers (contract_d's order of operations)
template <int NKT, int DK16, int PR = 0>
__device__ __forceinline__ void contract_d_regs(f32x4 (&acc)[2][NKT], const f32x4 (&af)[NKT][DK16], const f32x4 (&bf)[2][DK16]) {
  if constexpr (PR == 1) {
    s16x4 bp[2][DK16];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int j = 0; j < DK16; ++j) bp[tm][j] = bf16x4_rne(bf[tm][j]);
#pragma unroll
    for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
      for (int j = 0; j < DK16; ++j) {
        const s16x4 a = bf16x4_rne(af[tn][j]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm][tn] = mfma_bf16(a, bp[tm][j], acc[tm][tn]);
      }
    return;
  }
#pragma unroll
  for (int tn = 0; tn < NKT; ++tn)
#pragma unroll
    for (int j = 0; j < DK16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tn][j][e], bf[tm][j][e], acc[tm][tn], 0, 0, 0);
}
// the A operand of contract_q_acc for all 8 steps, requested in one batch; masked when it is about to be used
template <int DK16>
__device__ __forceinline__ void raw_q_cols(float (&af)[8][DK16], const float* __restrict__ A, int64_t lda, int n_q, int i, int g) {
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int q = st * 4 + g;
    Ld<DK16>::ld(A + (int64_t)(q < n_q ? q : n_q - 1) * lda + DK16 * i, af[st]);
  }
}
template <int DK16>
__device__ __forceinline__ void mask_q_cols(float (&af)[8][DK16], int n_q, int g) {
#pragma unroll
  for (int st = 0; st < 8; ++st)
#pragma unroll
    for (int td = 0; td < DK16; ++td) af[st][td] = (st * 4 + g < n_q) ? af[st][td] : 0.f;
}
template <int DK16, int PR = 0>
__device__ __forceinline__ void contract_q_acc_regs(const float* T, int pitch, const float (&af)[8][DK16],
                                                    f32x4 (&acc)[DK16][2], int i, int g) {
  if constexpr (PR == 1) {
#pragma unroll
    for (int st0 = 0; st0 < 8; st0 += 4) {
      s16x4 a[DK16];
#pragma unroll
      for (int td = 0; td < DK16; ++td) a[td] = bf16x4_rne(af[st0][td], af[st0 + 1][td], af[st0 + 2][td], af[st0 + 3][td]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c = t * 16 + i;
        const s16x4 b = bf16x4_rne(T[(st0 * 4 + g) * pitch + c], T[((st0 + 1) * 4 + g) * pitch + c],
                                   T[((st0 + 2) * 4 + g) * pitch + c], T[((st0 + 3) * 4 + g) * pitch + c]);
#pragma unroll
        for (int td = 0; td < DK16; ++td) acc[td][t] = mfma_bf16(a[td], b, acc[td][t]);
      }
    }
    return;
  }
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int q = st * 4 + g;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float b = T[q * pitch + t * 16 + i];
#pragma unroll
      for (int td = 0; td < DK16; ++td) acc[td][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st][td], b, acc[td][t], 0, 0, 0);
    }
  }
}

// (first form of the dK / dV contraction: operand loaded inside each step; bwd_kb_kernel<.., NB = 0>)
template <int DK16>
__device__ __forceinline__ void contract_q_acc(const float* T, int pitch, const float* __restrict__ A, int64_t lda,
                                               int n_q, f32x4 (&acc)[DK16][2], int i, int g) {
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int q = st * 4 + g;
    float af[DK16];
#pragma unroll
    for (int td = 0; td < DK16; ++td) af[td] = 0.f;
    if (q < n_q) Ld<DK16>::ld(A + (int64_t)q * lda + DK16 * i, af);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float b = T[q * pitch + t * 16 + i];
#pragma unroll
      for (int td = 0; td < DK16; ++td) acc[td][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[td], b, acc[td][t], 0, 0, 0);
    }
  }
}

template <int DK16>
__device__ __forceinline__ void store_key_rows(const f32x4 (&acc)[DK16][2], float* __restrict__ dst, int64_t ldd,
                                               int n_keys, int i, int g) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int key = t * 16 + i;
    if (key < n_keys) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float v[DK16];
#pragma unroll
        for (int td = 0; td < DK16; ++td) v[td] = acc[td][t][rr];
        Ld<DK16>::st(dst + (int64_t)key * ldd + DK16 * (g * 4 + rr), v);
      }
    }
  }
}

#ifndef NACF_ATTN_KB_WGS
#define NACF_ATTN_KB_WGS 2
#endif
constexpr int KB_WGS = NACF_ATTN_KB_WGS;        // resident workgroups per CU the register budget is set for
                                                // (measured: 1 -> 121 us, 2 -> 97 us, 3 spills 71 VGPRs -> 145 us)
constexpr int KB_PITCH = 48;                     // transpose tile [32][48]: 2 key tiles + 16 (rows 16 banks apart)
constexpr int KB_RED_PITCH = 68;                 // dQ partials [4][32][64 + 4]
template <int DK16, int NB = 1, int PR = 0>      // NB = 0: the guarded loads of the first form (kept for A/B: NACF_ATTN_KB=2; fp32 only)
__global__ __launch_bounds__(256, KB_WGS) void bwd_kb_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                      int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                                      const float* __restrict__ dO, int64_t lddo, float* __restrict__ dQ,
                                                      int64_t lddq, float* __restrict__ dK, int64_t lddk,
                                                      float* __restrict__ dV, int64_t lddv, int R, int n_kv, int H, int Lq,
                                                      int Lk, int kv_div, int kv_mod, int rounds) {
  constexpr int DK = 16 * DK16;
  static_assert(DK <= 64, "dQ reduce buffer is sized for dk <= 64");
  static_assert(NB == 1 || PR == 0, "the first form is fp32 only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ex = smem;                                             // [3][4][32]: block max, block sum, block delta
  float* red = ex + 3 * 4 * 32;                                 // [4][32][KB_RED_PITCH] dQ partials ...
  float* T = red + wave * 32 * KB_PITCH;                        // ... later reused as wave-private transpose tiles
  const int item = blockIdx.x;
  const int kvr = item / H, h = item % H;
  const int i = lane & 15, g = lane >> 4;
  const float sq = sqrtf((float)DK);
  const int k0 = wave * 32;
  const int nk = max(0, min(32, Lk - k0));                      // live keys of this wave's block
  const float* Kb = K + ((int64_t)kvr * Lk + k0) * ldk + h * DK;
  const float* Vb = V + ((int64_t)kvr * Lk + k0) * ldv + h * DK;
  f32x4 accK[DK16][2], accV[DK16][2];
#pragma unroll
  for (int td = 0; td < DK16; ++td)
#pragma unroll
    for (int t = 0; t < 2; ++t) { accK[td][t] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[td][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int k = 0; k < rounds; ++k) {
    // the k-th sequence of this memory row set (same enumeration as bwd_kernel); uniform over the workgroup
    const int r = ((k / kv_div) * kv_mod + kvr) * kv_div + k % kv_div;
    if (!(r < R && (k / kv_div) * kv_mod + kvr < (R + kv_div - 1) / kv_div)) continue;
    const float* Qb = Q + (int64_t)r * Lq * ldq + h * DK;
    const float* dOb = dO + (int64_t)r * Lq * lddo + h * DK;
    f32x4 p[2][2], dp[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int t = 0; t < 2; ++t) { p[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float bm[2] = {-3.0e38f, -3.0e38f}, bl[2] = {0.f, 0.f};
    if (nk > 0) {
      if constexpr (NB) {
        // each contraction's operands are requested as one batch; the zeroing of rows past the end follows the wait.  The scheduling
        // barriers keep a batch a batch (the scheduler sinks every load to its first use).  All four operands at once would hide the dP
        // operands under the S contraction, but needs more than the 256 registers two workgroups per CU leave a wave (148 B of spills).
        {
          f32x4 qf[2][DK16], kf[2][DK16];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            raw_row_frag<DK16>(qf[t], Qb, ldq, t * 16 + i, Lq, g);
            raw_row_frag<DK16>(kf[t], Kb, ldk, t * 16 + i, nk, g);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 2; ++t) { mask_row_frag<DK16>(qf[t], t * 16 + i < Lq); mask_row_frag<DK16>(kf[t], t * 16 + i < nk); }
          contract_d_regs<2, DK16, PR>(p, kf, qf);                      // S block
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          f32x4 gf[2][DK16], vf[2][DK16];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            raw_row_frag<DK16>(gf[t], dOb, lddo, t * 16 + i, Lq, g);
            raw_row_frag<DK16>(vf[t], Vb, ldv, t * 16 + i, nk, g);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 2; ++t) { mask_row_frag<DK16>(gf[t], t * 16 + i < Lq); mask_row_frag<DK16>(vf[t], t * 16 + i < nk); }
          contract_d_regs<2, DK16, PR>(dp, vf, gf);                     // dP block = dO V_blk^T
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        {
          f32x4 qf[2][DK16];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) load_row_frag<DK16>(qf[tm], Qb, ldq, tm * 16 + i, Lq, g);
          contract_d<2, DK16>(p, Kb, ldk, nk, qf, i, g);
        }
        {
          f32x4 gf[2][DK16];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) load_row_frag<DK16>(gf[tm], dOb, lddo, tm * 16 + i, Lq, g);
          contract_d<2, DK16>(dp, Vb, ldv, nk, gf, i, g);
        }
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int key = t * 16 + g * 4 + rr;
            const float v = key < nk ? p[tm][t][rr] / sq : -3.0e38f;
            p[tm][t][rr] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float e = expf(p[tm][t][rr] - mx);               // tile padding: exp(-3e38 - mx) = 0
            p[tm][t][rr] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        bm[tm] = mx;
        bl[tm] = sum;
      }
    }
    if (g == 0) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        ex[(0 * 4 + wave) * 32 + tm * 16 + i] = bm[tm];
        ex[(1 * 4 + wave) * 32 + tm * 16 + i] = bl[tm];
      }
    }
    __syncthreads();
    float part[2] = {0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int q = tm * 16 + i;
      float M = -3.0e38f;
#pragma unroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, ex[(0 * 4 + w) * 32 + q]);
      float Lsum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) Lsum += ex[(1 * 4 + w) * 32 + q] * expf(ex[(0 * 4 + w) * 32 + q] - M);
      const float scale = expf(bm[tm] - M) / Lsum;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          p[tm][t][rr] *= scale;                                   // P block
          part[tm] += p[tm][t][rr] * dp[tm][t][rr];
        }
      part[tm] += __shfl_xor(part[tm], 16, 64);
      part[tm] += __shfl_xor(part[tm], 32, 64);
    }
    if (g == 0) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) ex[(2 * 4 + wave) * 32 + tm * 16 + i] = part[tm];
    }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int q = tm * 16 + i;
      float dot = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) dot += ex[(2 * 4 + w) * 32 + q];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) dp[tm][t][rr] = p[tm][t][rr] * (dp[tm][t][rr] - dot) / sq;   // dS block
    }
    {
      f32x4 o[2][DK16];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int td = 0; td < DK16; ++td) o[tm][td] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nk > 0) contract_key<2, DK16, PR>(o, dp, Kb, ldk, nk, i, g);   // dQ partial = dS_blk K_blk
      float* mine = red + wave * 32 * KB_RED_PITCH;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float v[DK16];
#pragma unroll
          for (int td = 0; td < DK16; ++td) v[td] = o[tm][td][rr];
          Ld<DK16>::st(mine + (tm * 16 + i) * KB_RED_PITCH + DK16 * (g * 4 + rr), v);
        }
    }
    __syncthreads();
    {
      // 32 rows x DK columns summed over the 4 key blocks in wave order; thread -> (row, 8 columns)
      constexpr int CPT = DK / 8;                                   // threads per row
      const int q = threadIdx.x / CPT, d0 = (threadIdx.x % CPT) * 8;
      if (q < 32 && q < Lq) {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float* src = red + (w * 32 + q) * KB_RED_PITCH + d0;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
          a0 += x0; a1 += x1;
        }
        float* out = dQ + ((int64_t)r * Lq + q) * lddq + h * DK + d0;
        *reinterpret_cast<f32x4*>(out) = a0;
        *reinterpret_cast<f32x4*>(out + 4) = a1;
      }
    }
    __syncthreads();                                            // the transpose tiles alias the dQ partials
    if (nk > 0) {
      if constexpr (NB) {
        // (requested two barriers earlier, ahead of the dQ reduction, these 64 registers spill: 97 -> 114 us; dQ's alone: 102 us)
        float aq[8][DK16], ao[8][DK16];
        raw_q_cols<DK16>(aq, Qb, ldq, Lq, i, g);
        raw_q_cols<DK16>(ao, dOb, lddo, Lq, i, g);
        __builtin_amdgcn_sched_barrier(0);
        mask_q_cols<DK16>(aq, Lq, g);
        tile_to_lds<2>(T, KB_PITCH, dp, i, g);
        contract_q_acc_regs<DK16, PR>(T, KB_PITCH, aq, accK, i, g);        // dK_blk += dS_blk^T Q
        __builtin_amdgcn_sched_barrier(0);                             // (dO's wait and zeroing stay behind the dK contraction)
        mask_q_cols<DK16>(ao, Lq, g);
        tile_to_lds<2>(T, KB_PITCH, p, i, g);
        contract_q_acc_regs<DK16, PR>(T, KB_PITCH, ao, accV, i, g);        // dV_blk += P_blk^T dO
      } else {
        tile_to_lds<2>(T, KB_PITCH, dp, i, g);
        contract_q_acc<DK16>(T, KB_PITCH, Qb, ldq, Lq, accK, i, g);
        tile_to_lds<2>(T, KB_PITCH, p, i, g);
        contract_q_acc<DK16>(T, KB_PITCH, dOb, lddo, Lq, accV, i, g);
      }
    }
  }
  if (nk > 0) {
    store_key_rows<DK16>(accK, dK + ((int64_t)kvr * Lk + k0) * lddk + h * DK, lddk, nk, i, g);
    store_key_rows<DK16>(accV, dV + ((int64_t)kvr * Lk + k0) * lddv + h * DK, lddv, nk, i, g);
  }
}

// ---- LDS-staged forward (cross-attention: no key mask, not causal, dk = 64, up to 128 keys) ---------------------------
// fwd_kernel streams K and V straight from L2 into MFMA fragments, once per wave: every (sequence, head, block of 32
// queries) re-reads the same 61 KB, and with two waves per SIMD nothing covers the load latency (80 us for the decode
// cross-attention whose MFMA time is ~16 us).  Here ONE workgroup owns a (memory row set, head): it copies K and V into
// LDS once with wide coalesced loads (row pitch 68 floats), and its four waves then walk the (sequence, query block)
// items of that row set -- the 6 length-beam candidates of a video as 4 blocks of 32 queries, or the two NACF passes
// of a video -- taking every MFMA operand from LDS.  Same contractions, same softmax, same accumulation order per
// query row as fwd_kernel: the results are bit-identical.
constexpr int FL_PITCH = 68;
// the work of one workgroup for one (memory row set kvr, head h); smem: 2 x 128 x FL_PITCH floats (a caller that walks
// several items puts a barrier before the next copy).
template <int DK16, int PR = 0>
__device__ __forceinline__ void fwd_lds_item(const int item, float* const smem, const float* __restrict__ Q, int64_t ldq,
                                             const float* __restrict__ K, int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                             float* __restrict__ O, int64_t ldo, int R, int H, int Lq, int Lk, int kv_div,
                                             int kv_mod, int rounds) {
  constexpr int DK = 16 * DK16;
  constexpr int NKT = 8;
  float* Ks = smem;                               // [128][FL_PITCH], rows >= Lk are never read as live keys
  float* Vs = smem + 128 * FL_PITCH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kvr = item / H, h = item % H;
  const int i = lane & 15, g = lane >> 4;
  {
    const float* Kg = K + (int64_t)kvr * Lk * ldk + h * DK;
    const float* Vg = V + (int64_t)kvr * Lk * ldv + h * DK;
    constexpr int VPR = DK / 4;                   // f32x4 per row
    constexpr int NLD = 128 * VPR / 256;          // loads per thread and operand: ALL issued before the first LDS store,
    f32x4 kr[NLD], vr[NLD];                       // so the copy costs one trip to L2, not NLD of them
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = threadIdx.x + 256 * u;
      const int row = min(q / VPR, Lk - 1), c = (q % VPR) * 4;
      kr[u] = *reinterpret_cast<const f32x4*>(Kg + (int64_t)row * ldk + c);
      vr[u] = *reinterpret_cast<const f32x4*>(Vg + (int64_t)row * ldv + c);
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int q = threadIdx.x + 256 * u;
      const int row = q / VPR, c = (q % VPR) * 4;
      if (row < Lk) {
        *reinterpret_cast<f32x4*>(&Ks[row * FL_PITCH + c]) = kr[u];
        *reinterpret_cast<f32x4*>(&Vs[row * FL_PITCH + c]) = vr[u];
      }
    }
  }
  __syncthreads();
  const int nqb = (Lq + 31) / 32;
  const int n_items = rounds * nqb;               // (k-th sequence of the row set, block of 32 queries)
  const float sq = sqrtf((float)DK);
  for (int it = wave; it < n_items; it += 4) {
    const int k = it / nqb, qb = it % nqb;
    const int r = ((k / kv_div) * kv_mod + kvr) * kv_div + k % kv_div;
    if (!(r < R && (k / kv_div) * kv_mod + kvr < (R + kv_div - 1) / kv_div)) continue;
    const int q0 = qb * 32;
    const int nq = min(32, Lq - q0);
    const float* Qb = Q + ((int64_t)r * Lq + q0) * ldq + h * DK;
    f32x4 qf[2][DK16];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) load_row_frag<DK16>(qf[tm], Qb, ldq, tm * 16 + i, nq, g);
    f32x4 sc[2][NKT];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < NKT; ++tn) sc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    contract_d<NKT, DK16, PR>(sc, Ks, FL_PITCH, Lk, qf, i, g);
    softmax_rows<NKT>(sc, sq, nullptr, 0, Lk, i, g);
    f32x4 o[2][DK16];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int td = 0; td < DK16; ++td) o[tm][td] = f32x4{0.f, 0.f, 0.f, 0.f};
    contract_key<NKT, DK16, PR>(o, sc, Vs, FL_PITCH, Lk, i, g);
    store_rows<DK16>(o, O + ((int64_t)r * Lq + q0) * ldo + h * DK, ldo, nq, i, g);
  }
}

template <int DK16, int PR = 0>
__global__ __launch_bounds__(256, 2) void fwd_lds_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                          int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                                          float* __restrict__ O, int64_t ldo, int R, int n_kv, int H, int Lq,
                                                          int Lk, int kv_div, int kv_mod, int rounds) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  fwd_lds_item<DK16, PR>((int)blockIdx.x, smem, Q, ldq, K, ldk, V, ldv, O, ldo, R, H, Lq, Lk, kv_div, kv_mod, rounds);
}

}  // namespace attn
